"""CPU, world_size 2 over gloo: the data-parallel path (dist.GradReducer) averages gradients
across ranks -- both the overlapped arena-bucket path driven by engine callbacks and the
coalesced path for plug-in modules -- and broadcast_parameters aligns the replicas."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import pai_bootstrap
    pai_bootstrap.load()
    from thesis_pai_reconstruction_amd import dist as pdist, engine as E
    r, _, w = pdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    try:
        # ---- arena path: a fake engine that reports progress like UnetEngine.backward does -------
        torch.manual_seed(0)
        convs = [nn.Conv2d(8, 16, 4, 2, 1), nn.ConvTranspose2d(16, 8, 4, 2, 1), nn.Conv2d(1, 64, 4, 2, 1)]
        for c in convs:
            E.to_fwd_pack_(c)
        entries = []
        for c in convs:
            entries += [(c.weight, c), (c.bias, None)]

        class FakeEngine:
            grad_ready_hook = None

            def __init__(self):
                self._arena = E.GradArena(entries, torch.device("cpu"))

            def arena(self):
                return self._arena

            def ordered_params(self):
                return entries

        eng = FakeEngine()
        red = pdist.GradReducer(bucket_bytes=4096, overlap=True)
        red.attach_engine(eng)
        params = [p for p, _ in entries]
        for step in range(2):
            A = eng.arena()
            A.begin_backward(params) if step == 0 else A.flat.zero_()
            for p in params:                       # "backward": each rank writes rank-dependent grads
                A.seg(p).copy_(torch.full((p.numel(),), float(rank + 1 + step)))
                eng.grad_ready_hook(A, A.end_of(p))
            A.attach(params)
            red.finish()
            want = sum(float(k + 1 + step) for k in range(world)) / world
            for p in params:
                assert torch.allclose(p.grad, torch.full_like(p.grad, want)), (rank, step)
        assert red.stats["buckets"] >= 4           # several buckets were launched before finish()
        # a second backward into an arena whose buckets were already sent must be refused
        A = eng.arena()
        eng.grad_ready_hook(A, A.end_of(params[-1]))
        with pytest.raises(RuntimeError):
            eng.grad_ready_hook(A, A.end_of(params[0]))
        red.finish()

        # ---- bf16 wire format: same buckets, half the bytes, bf16 rounding of the summands; fp32 master gradients ----
        eng16 = FakeEngine()
        red16 = pdist.GradReducer(bucket_bytes=4096, overlap=True, grad_dtype=torch.bfloat16)
        red16.attach_engine(eng16)
        A = eng16.arena()
        for p in params:
            p.grad = None                           # the views of the first engine's arena
        A.begin_backward(params)
        gen = torch.Generator().manual_seed(1234 + rank)
        local = {}
        for p in params:
            g = torch.randn(p.numel(), generator=gen)
            local[id(p)] = g
            A.seg(p).copy_(g)
            eng16.grad_ready_hook(A, A.end_of(p))
        A.attach(params)
        red16.finish()
        assert red16.stats["bytes"] * 2 == A.flat.numel() * 4 and A.flat.dtype == torch.float32
        for p in params:
            both = [torch.zeros_like(local[id(p)]) for _ in range(world)]
            dist.all_gather(both, local[id(p)])
            want32 = sum(both) / world                                     # what the fp32 buckets give
            want16 = sum(b.bfloat16() for b in both).float() / world       # summands and sum rounded to bf16
            got = A.seg(p)
            assert torch.allclose(got, want16, atol=1e-6), (rank, float((got - want16).abs().max()))
            assert float((got - want32).abs().max()) <= 2 ** -7 * float(sum(b.abs() for b in both).max()) + 1e-6

        # ---- plug-in module path + parameter broadcast ---------------------------------------------
        torch.manual_seed(100 + rank)              # ranks start from DIFFERENT weights
        net = nn.Sequential(nn.Conv2d(1, 4, 3, padding=1), nn.ReLU(), nn.Conv2d(4, 1, 3, padding=1))
        E.to_fwd_pack_(nn.Conv2d(1, 1, 4))         # (no-op on an unrelated module)
        pdist.broadcast_parameters(net, src=0)
        flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert all(torch.equal(gathered[0], g) for g in gathered)
        red2 = pdist.GradReducer()
        red2.attach(net)
        x = torch.full((2, 1, 8, 8), float(rank + 1))
        net(x).sum().backward()
        local = [p.grad.clone() for p in net.parameters()]
        red2.finish()
        for p, g in zip(net.parameters(), local):
            both = [torch.zeros_like(g) for _ in range(world)]
            dist.all_gather(both, g)
            assert torch.allclose(p.grad, sum(both) / world, atol=1e-6)
        # small buckets: the 144-element conv weight goes in place, the rest in coalesced buckets of >= 16 elements
        red3 = pdist.GradReducer(bucket_bytes=64)
        red3.attach(net)
        net.zero_grad(set_to_none=True)
        nn.Sequential(net, nn.Conv2d(1, 1, 1))      # (unrelated container; net's parameters are what is attached)
        wide = nn.Conv2d(4, 4, 3, padding=1)
        net2 = nn.Sequential(net[0], nn.ReLU(), wide, nn.ReLU(), net[2])
        pdist.broadcast_parameters(net2, src=0)
        # the local gradients come from a twin without hooks: tensors of at least one bucket are all-reduced IN PLACE by the
        # hook the moment they are final, i.e. while the backward pass is still running
        import copy
        twin = copy.deepcopy(net2)
        twin(x).sum().backward()
        local = [p.grad.clone() for p in twin.parameters()]
        red3.attach(net2)
        net2(x).sum().backward()
        assert red3.stats["buckets"] >= 3 and red3.stats["foreign_buckets"] >= 3     # issued before finish()
        assert red3.plannable()
        red3.finish()
        assert red3.stats["buckets"] >= 4
        for p, g in zip(net2.parameters(), local):
            both = [torch.zeros_like(g) for _ in range(world)]
            dist.all_gather(both, g)
            assert torch.allclose(p.grad, sum(both) / world, atol=1e-6)
        # the same buckets with the bf16 wire format, and with the overlap off (everything issued in finish()): both must
        # average to the fp32 result within bf16 rounding / exactly
        for kwargs, tol in (({"grad_dtype": torch.bfloat16}, 2 ** -7), ({"overlap": False}, 1e-6)):
            net3 = copy.deepcopy(twin)
            net3.zero_grad(set_to_none=True)
            red4 = pdist.GradReducer(bucket_bytes=64, **kwargs)
            red4.attach(net3)
            net3(x).sum().backward()
            if not red4.overlap:
                assert red4.stats["buckets"] == 0
            red4.finish()
            assert red4.stats["buckets"] >= 4
            for p, g in zip(net3.parameters(), local):
                both = [torch.zeros_like(g) for _ in range(world)]
                dist.all_gather(both, g)
                want = sum(both) / world
                assert float((p.grad - want).abs().max()) <= tol * max(1.0, float(sum(b.abs() for b in both).max())), kwargs
        # permuted (fwd-pack) parameters are broadcast through their storage order
        ct = nn.ConvTranspose2d(8, 4, 4, 2, 1)
        E.to_fwd_pack_(ct)
        pdist.broadcast_parameters(ct, src=0)
        both = [torch.zeros(ct.weight.numel()) for _ in range(world)]
        dist.all_gather(both, ct.weight.detach().reshape(-1).contiguous())
        assert torch.equal(both[0], both[1])
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_grad_reducer_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", f"rank {rank}:\n{msg}"
