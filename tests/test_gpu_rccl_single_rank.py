"""GPU: the data-parallel machinery on the REAL backend of the multi-GPU launch -- torch.distributed
"nccl" = RCCL -- in a one-rank group (all a one-GPU box can host; FI_DP_FORCE=1 makes the engine issue its
collectives although world_size is 1).  What this exercises before an 8-GPU node ever sees it: process-group
initialisation with a bound device, the bucketed async all-reduce on the side HIP stream with work handles
and stream hand-over, the statistics all-reduce inside autograd, broadcast_parameters, the per-bucket
consistency flags.  A one-rank all-reduce is the identity, so the step must reproduce the plain
single-process step bit for bit (same kernels, same order on the main stream)."""
import os
import socket
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(use_dp, out):
    import torch.distributed as dist
    from feature_intertwiner_amd.config import make_config
    from feature_intertwiner_amd.data_parallel import GradientBuckets, all_reduce_statistics, broadcast_parameters
    from feature_intertwiner_amd.model import MaskRCNN
    from feature_intertwiner_amd.synthetic import SyntheticProposals, synthetic_batch
    from feature_intertwiner_amd.workflow import set_optimizer, train_step
    torch.manual_seed(99)
    cfg = make_config("resnet50", 256, 2, 64, dev_switch=True, loss_choice="l2", loss_fac=50.0)
    model = MaskRCNN(cfg).to(DEV)
    sync = reduce_fn = None
    if use_dp:
        broadcast_parameters(model)
        sync = GradientBuckets(model, bucket_bytes=8 << 20)
        assert sync.active and len(sync.buckets) > 3
        reduce_fn = all_reduce_statistics
    opt = set_optimizer(model, cfg.TRAIN)
    batch = synthetic_batch(2, 256, device=DEV)
    model.external_proposals = SyntheticProposals(batch[2], 256, seed=7)
    model.generator = torch.Generator(device=DEV).manual_seed(5)
    terms = train_step(model, opt, list(batch), grad_sync=sync, world_size=1, reduce_fn=reduce_fn)
    params = [p.detach().cpu().clone() for p in model.parameters()]          # after ONE update
    for _ in range(2):            # later steps: the engine's per-step state must reset (not compared: discrete
        t2 = train_step(model, opt, list(batch), grad_sync=sync, world_size=1, reduce_fn=reduce_fn)   # sampling amplifies last-bit drift)
        assert all(torch.isfinite(v) for v in t2.values())
    torch.cuda.synchronize()
    if sync is not None:
        sync.check()
        # after the first real collectives: the measured concurrency of the picked streams (second, third, communication),
        # a stream that lost it is replaced -- and the engine goes on working on whatever it ended up with
        choice = sync.recheck_streams()
        roles = [r["role"] for r in choice["streams"]]
        assert "communication" in roles and "second" in roles, roles
        assert all(isinstance(r["concurrent_with_earlier"], bool) for r in choice["streams"])
        assert sum(1 for r in choice["streams"] if r["concurrent_with_earlier"]) >= 2, choice     # 4 hardware queues: main + 2 at least
        t3 = train_step(model, opt, list(batch), grad_sync=sync, world_size=1, reduce_fn=reduce_fn)
        assert all(torch.isfinite(v) for v in t3.values()) and sync.late_gradients == 0
        torch.cuda.synchronize()
        sync.check()
    torch.save({"terms": {k: float(v) for k, v in terms.items()}, "params": params}, out)


def _worker(rank, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["FI_DP_FORCE"] = "1"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        _run(True, out)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_rccl_one_rank_step_is_the_single_process_step():
    with tempfile.TemporaryDirectory() as d:
        dp, plain = os.path.join(d, "dp.pt"), os.path.join(d, "plain.pt")
        mp.spawn(_worker, args=(_free_port(), dp), nprocs=1, join=True)
        _run(False, plain)
        a, b = torch.load(dp, weights_only=False), torch.load(plain, weights_only=False)
    for k in b["terms"]:          # first step: same weights, same inputs
        assert abs(a["terms"][k] - b["terms"][k]) <= 1e-6 * max(1.0, abs(b["terms"][k])), (k, a["terms"][k], b["terms"][k])
    worst = 0.0
    for pa, pb in zip(a["params"], b["params"]):
        worst = max(worst, (pa - pb).abs().max().item() / (pb.abs().max().item() + 1e-12))
    # two runs of the same step differ in the last bits of the gradients (fp32 atomics in the weight-gradient
    # splits and the RoIAlign backward)
    assert worst <= 1e-4, worst
