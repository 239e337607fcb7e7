"""Data-parallel path on ONE GPU with TWO processes (gloo process group on device tensors, both ranks on cuda:0) - what can
be verified of BASELINE.json configs[2] without an 8-GPU node (reference README.md:45-57 ``--num_gpus/--gpus_per_net``,
README.md:756-760 ``cfg.gpus``):

* a 2-rank step with the overlapped gradient all-reduce == one process accumulating the two half-batch gradients, bit for bit
  in fp32 mode (per-replica BatchNormalization statistics, 1/world folded into the optimizer);
* a 2-rank ``cfg.fit()`` with EarlyStopping / ReduceLROnPlateau, sharded validation and a short last batch ends on every rank
  at the same epoch with identical parameters (no rank-divergent control flow, no hang)."""
import csv
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import yaml

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_step_equals_accumulated_half_batches(tmp_path):
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   STP_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), str(tmp_path)], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    p0, p1, ref = (np.load(str(tmp_path / n)) for n in ("P_rank0.npy", "P_rank1.npy", "P_ref.npy"))
    assert open(str(tmp_path / "ok_rank0")).read() == "1" and open(str(tmp_path / "ok_rank1")).read() == "1"
    assert np.array_equal(p0, p1)                                   # the replicas stay identical
    assert np.array_equal(p0, ref), float(np.abs(p0 - ref).max())   # == accumulated half batches, bitwise (fp32 mode)


def test_two_rank_fit_is_rank_consistent(tmp_path):
    from test_fit_gpu import make_dataset
    make_dataset(str(tmp_path), n=11)                                # 11 samples: shards and last batches are ragged
    cfg_path = str(tmp_path / "config.yaml")
    with open(cfg_path, "w") as f:
        yaml.safe_dump({"architecture": "Unet", "backbone": "resnet18", "classes": 1, "activation": "sigmoid", "encoder_weights": None,
                        "shape": [64, 64, 3], "optimizer": "Adam", "lr": 0.01, "batch": 2, "folds_count": 2, "gpus": 2, "dtype": "fp32",
                        "loss": "binary_crossentropy+1.0*dice_loss", "metrics": ["binary_accuracy", "dice"],
                        "primary_metric": "val_loss", "draw_examples": False,
                        "fit_with": "train", "datasets": {"train": {"input_path": "train", "output_path": "train_mask"}},
                        # patience 1 on a noisy tiny run: ranks deciding on their OWN shard would stop at different epochs
                        "callbacks": {"EarlyStopping": {"patience": 1, "monitor": "val_loss"},
                                      "ReduceLROnPlateau": {"patience": 1, "factor": 0.5, "monitor": "loss"}},
                        "stages": [{"epochs": 12}, {"epochs": 2, "lr": 0.001}]}, f)
    env = dict(os.environ, STP_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    code = ("import sys; sys.path.insert(0, %r); from segmentation_pipeline import segmentation; "
            "cfg = segmentation.parse(%r); out = cfg.fit(foldsToExecute=[0]); print('STAGES', len(out))" % (ROOT, cfg_path))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)   # a hang fails here
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "STAGES 2" in r.stdout
    with open(str(tmp_path / "summary.yaml")) as f:
        summ = yaml.safe_load(f)["stages"]
    assert [(s["fold"], s["stage"]) for s in summ] == [(0, 0), (0, 1)]
    with open(str(tmp_path / "metrics" / "metrics-0.0.csv")) as f:
        rows = list(csv.DictReader(f))
    assert len(rows) == summ[0]["epochs_run"] and np.all(np.isfinite([float(r_["val_loss"]) for r_ in rows]))
    assert os.path.exists(str(tmp_path / "weights" / "best-0.1.weights"))


def test_bench_two_ranks_on_one_gpu_prints_one_line():
    """The driver's multi-GPU form of bench.py, dry-run with two ranks on ONE GPU (gloo instead of RCCL, which refuses duplicate
    devices; the rank -> device map wraps around): exactly one JSON line, from rank 0, with the world size in it - so the first
    8-GPU run is not the first time ``bench.py --gpus N`` executes."""
    import json
    env = dict(os.environ, STP_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-kernel-profile"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["config"]["global_batch"] == 32
    assert out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak" and out["value"] > 0
    assert "cpu_baseline" not in out and "roofline" not in out          # rank 0 at N = 1 only / --no-kernel-profile
