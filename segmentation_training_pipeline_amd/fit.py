"""``musket fit`` for the HIP backend: launches experiment training, one process per GPU.

Mirrors the launcher flags of the reference (README.md:37-57): ``--project`` (project root), ``--name`` (experiment
sub-folder(s) holding ``config.yaml``, comma separated; default: every experiment of the project), ``--num_gpus`` (GPUs
this launch may use), ``--gpus_per_net`` (GPUs per experiment = data-parallel world size), ``--allow_resume``,
``--folds``.  The reference replicates the Keras graph inside one process (``multi_gpu_model``); here every experiment
becomes ``gpus_per_net`` ranks started through ``torch.distributed.run`` (RCCL over xGMI, rendezvous on 127.0.0.1), and
``num_gpus // gpus_per_net`` experiments run side by side on disjoint GPU sets (``HIP_VISIBLE_DEVICES``).

    python -m segmentation_training_pipeline_amd.fit --project path/to/project --name exp1 --num_gpus 8 --gpus_per_net 8

A worker (``--worker``) parses the YAML and calls ``cfg.fit()`` with the dataset declared in it (``fit_with`` /
``datasets``, examples/people/ds_1.yaml:33-38).
"""
import argparse
import os
import subprocess
import sys


def find_experiments(project, names=None):
    """[(name, path to config.yaml)]: ``<project>/experiments/<name>/config.yaml`` (musket project layout), falling back to
    ``<project>/<name>/config.yaml`` and to ``<project>/<name>`` being the YAML itself."""
    roots = [os.path.join(project, "experiments"), project]
    if not names:
        names = []
        for r in roots:
            if os.path.isdir(r):
                names = sorted(d for d in os.listdir(r) if os.path.isfile(os.path.join(r, d, "config.yaml")))
                if names:
                    break
    out = []
    for n in names:
        cands = [os.path.join(r, n, "config.yaml") for r in roots] + [os.path.join(project, n)]
        hit = next((c for c in cands if os.path.isfile(c)), None)
        if hit is None:
            raise FileNotFoundError("experiment %r: no config.yaml under %s" % (n, project))
        out.append((n, hit))
    return out


def plan_launches(experiments, num_gpus, gpus_per_net, base_port=29500):
    """Waves of concurrently running experiments: [[{name, config, devices, nproc, port}, ...], ...].  Each experiment
    gets ``gpus_per_net`` consecutive devices; ``num_gpus // gpus_per_net`` of them run at a time."""
    num_gpus, gpus_per_net = max(1, int(num_gpus)), max(1, int(gpus_per_net))
    if gpus_per_net > num_gpus:
        raise ValueError("--gpus_per_net (%d) exceeds --num_gpus (%d)" % (gpus_per_net, num_gpus))
    slots = num_gpus // gpus_per_net
    waves = []
    for i, (name, cfg) in enumerate(experiments):
        if i % slots == 0:
            waves.append([])
        k = i % slots
        waves[-1].append({"name": name, "config": cfg, "nproc": gpus_per_net, "port": base_port + i,
                          "devices": list(range(k * gpus_per_net, (k + 1) * gpus_per_net))})
    return waves


def command(job, allow_resume=False, folds=None):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(job["nproc"]),
           "--master-addr", "127.0.0.1", "--master-port", str(job["port"]), "-m", "segmentation_training_pipeline_amd.fit",
           "--worker", job["config"]]
    if allow_resume:
        cmd.append("--allow_resume")
    if folds:
        cmd += ["--folds", ",".join(str(f) for f in folds)]
    return cmd


def worker(config, allow_resume, folds):
    from segmentation_pipeline import segmentation
    cfg = segmentation.parse(config)
    if allow_resume:
        cfg.setAllowResume(True)
    out = cfg.fit(foldsToExecute=folds)
    if int(os.environ.get("RANK", "0")) == 0:
        print("fit: %s -> %d (fold, stage) runs" % (config, len(out)))


def main(argv=None):
    ap = argparse.ArgumentParser(prog="fit", description=__doc__.split("\n")[0])
    ap.add_argument("--project", default=".")
    ap.add_argument("--name", default=None, help="experiment name(s), comma separated")
    ap.add_argument("--num_gpus", type=int, default=1)
    ap.add_argument("--gpus_per_net", type=int, default=1)
    ap.add_argument("--num_workers", type=int, default=1, help="accepted for compatibility: loading runs in a thread per rank")
    ap.add_argument("--cache", default=None, help="accepted for compatibility (no on-disk cache is needed)")
    ap.add_argument("--allow_resume", action="store_true")
    ap.add_argument("--folds", default=None, help="comma separated fold numbers")
    ap.add_argument("--worker", default=None, help=argparse.SUPPRESS)
    a = ap.parse_args(argv)
    folds = [int(f) for f in a.folds.split(",")] if a.folds else None
    if a.worker:
        return worker(a.worker, a.allow_resume, folds)
    exps = find_experiments(a.project, a.name.split(",") if a.name else None)
    if not exps:
        raise SystemExit("no experiments found under %s" % a.project)
    rc = 0
    for wave in plan_launches(exps, a.num_gpus, a.gpus_per_net):
        procs = []
        for job in wave:
            env = dict(os.environ, HIP_VISIBLE_DEVICES=",".join(str(d) for d in job["devices"]), HSA_ENABLE_IPC_MODE_LEGACY="0")
            procs.append((job, subprocess.Popen(command(job, a.allow_resume, folds), env=env)))
        for job, p in procs:
            r = p.wait()
            if r:
                print("experiment %s failed (exit %d)" % (job["name"], r), file=sys.stderr)
                rc = rc or r
    return rc


if __name__ == "__main__":
    sys.exit(main() or 0)
