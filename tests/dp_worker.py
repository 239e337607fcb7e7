"""One rank of the 2-process data-parallel equivalence test (tests/test_dp_gpu.py): both ranks share cuda:0, the process
group is gloo on device tensors.  Trains ONE step on this rank's half of a fixed batch with the overlapped gradient
all-reduce and writes its parameters; rank 0 additionally computes the single-process reference: the two half-batch
gradients accumulated (per-replica BatchNormalization semantics) and one optimizer step at 1/2 scale."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(graph):
    from segmentation_training_pipeline_amd.backend import HipSegModel
    return HipSegModel("Unet", "resnet18", (64, 64, 3), 1, "sigmoid", batch=2, dtype="fp32", loss="binary_crossentropy+1.0*dice_loss",
                       optimizer="Adam", lr=1e-3, use_graph=graph, device="cuda:0", seed=7)


def main(out_dir):
    from segmentation_training_pipeline_amd import distributed
    from oracle import step as ostep
    rank, _, world = distributed.init("gloo")
    x, y = ostep.synthetic_batch(4, 64, 64, seed=99)
    halves = [(x[:2], y[:2]), (x[2:], y[2:])]
    m = build(graph=True)
    m.broadcast_state(src=0)
    m.set_data_parallel(distributed.GradReducer(), overlap=True)
    # start-up measurement of the two schedules (backend.calibrate_dp_schedule): both run, every rank takes the same decision, and
    # the state is restored bit for bit - the reference comparison below would catch a stray optimizer step or BatchNormalization update
    m.load_batch(*halves[rank])
    before = [t.clone() for t in m._mutable_state()]
    rec = m.calibrate_dp_schedule(steps=2, warm=1, log=False)
    assert rec["chosen"] in ("overlapped", "serialised") and rec["overlapped_ms"] > 0 and rec["serialised_ms"] > 0, rec
    assert all(torch.equal(a, b) for a, b in zip(before, m._mutable_state())), "calibration must leave the state untouched"
    both = distributed.allreduce_sums([1.0 if rec["chosen"] == "overlapped" else 0.0])
    assert float(both[0]) in (0.0, 2.0), "the ranks disagree on the schedule"
    m.dp_overlap, m._segments, m._graphs = True, None, None          # the rest of the test exercises the overlapped schedule
    assert m._dp_segments() is not None and len(m._dp_segments()) >= 2, "the overlapped schedule must be active"
    for _ in range(2):                                        # two steps: the second one runs on replayed graphs
        m.train_on_batch(*halves[rank])
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, "P_rank%d.npy" % rank), m.plan.P.cpu().numpy())
    ok = distributed.replicas_equal(m.plan.P)
    if rank == 0:
        r = build(graph=False)
        r.gscale.fill_(0.5)
        r._build_opt(use_gscale=True)
        p = r.plan
        for _ in range(2):
            g = []
            for hx, hy in halves:
                r.load_batch(hx, hy)
                r.forward_backward()
                g.append(p.G.clone())
            p.G.copy_(g[0] + g[1])
            r.apply_gradients()
        torch.cuda.synchronize()
        np.save(os.path.join(out_dir, "P_ref.npy"), p.P.cpu().numpy())
    with open(os.path.join(out_dir, "ok_rank%d" % rank), "w") as f:
        f.write("1" if ok else "0")
    distributed.barrier()


if __name__ == "__main__":
    main(sys.argv[1])
