cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=${1:-r05e}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
printf 'group_off STP_STATS_GROUP=0\nmaxg2 STP_STATS_GROUP_MAXG=2\nmaxg4 STP_STATS_GROUP_MAXG=4\nmaxg16 STP_STATS_GROUP_MAXG=16\nmaxg2_noigemm STP_STATS_GROUP_MAXG=2 STP_STATS_GROUP_IGEMM=0\nmaxg16_noigemm STP_STATS_GROUP_MAXG=16 STP_STATS_GROUP_IGEMM=0\n' | bash scratch/r05/ab.sh $T
STP_STATS_GROUP=0 python scratch/launch_table.py > $O/launch_table_off.txt 2>&1
STP_STATS_GROUP_MAXG=16 python scratch/launch_table.py > $O/launch_table_g16.txt 2>&1
