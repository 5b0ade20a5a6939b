"""CPU: ``bench.launch_ranks`` -- what ``python bench.py --gpus N`` does when no launcher started it -- and the
precision of the cross-rank gradient sum (bf16 vs fp32 exchange), both over gloo.

The first multi-GPU run of ``bench.py`` must not be able to fail on plumbing: the launcher picks a free port, starts
``torch.distributed.run`` as a CHILD process (the parent never touches the GPU), hands rank 0's JSON line through and
returns the child's exit code.  The child here is a trivial script that initialises the process group (gloo) and lets
rank 0 print one JSON line -- the same hand-shake ``bench.py`` makes, without a GPU."""
import io
import json
import os
import socket
import sys
import textwrap

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import bench

CHILD_OK = textwrap.dedent("""
    import json, os, sys
    import torch, torch.distributed as dist
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    t = torch.tensor([float(os.environ["RANK"]) + 1.0])
    dist.all_reduce(t)
    dist.barrier()
    if dist.get_rank() == 0:
        print(json.dumps({"world": dist.get_world_size(), "sum": t.item(), "master": os.environ["MASTER_ADDR"],
                          "port": int(os.environ["MASTER_PORT"]), "argv": sys.argv[1:],
                          "ipc": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}), flush=True)
    dist.destroy_process_group()
""")

CHILD_FAIL = textwrap.dedent("""
    import os, sys
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    if int(os.environ["RANK"]) == int(os.environ["WORLD_SIZE"]) - 1:
        sys.exit(7)                       # one rank dies: the launcher must tear the job down and report non-zero
    dist.barrier()
""")


@pytest.mark.parametrize("world", [2, 4])
def test_launch_ranks_hands_back_rank0_json(tmp_path, world):
    child = tmp_path / "child.py"
    child.write_text(CHILD_OK)
    out = io.StringIO()
    rc = bench.launch_ranks([str(child), "--gpus", str(world), "--steps", "3"], world, stdout=out, timeout=240)
    assert rc == 0
    lines = [l for l in out.getvalue().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.getvalue()
    rec = json.loads(lines[0])
    assert rec["world"] == world and rec["sum"] == world * (world + 1) / 2
    assert rec["master"] == "127.0.0.1" and rec["port"] > 0 and rec["ipc"] == "0"
    assert rec["argv"] == ["--gpus", str(world), "--steps", "3"]       # the child's own arguments arrive unchanged


def test_launch_ranks_reports_a_failing_rank(tmp_path):
    child = tmp_path / "child.py"
    child.write_text(CHILD_FAIL)
    rc = bench.launch_ranks([str(child)], 2, stdout=io.StringIO(), timeout=240)
    assert rc != 0


def test_free_port_is_bindable():
    p = bench.free_port()
    s = socket.socket()
    s.bind(("127.0.0.1", p))
    s.close()


def test_defaults_and_self_launch_decision(monkeypatch):
    """--gpus N without RANK / WORLD_SIZE in the environment takes the self-launch branch (no SystemExit asking for a
    launcher any more); the contract line for N > 1 is the strong-scaling partition of SURVEY 8(e)."""
    a = bench.parse(["--gpus", "8"])
    assert a.gpus == 8 and a.scaling is None and a.dtype == "bf16" and a.workload == "literal"
    assert bench.parse(["--workload", "mnist_mlp"]).dtype == "f32"          # configs #2 / #3: fp32
    assert a.cpu_warmup == 3 and a.cpu_steps == 10
    called = {}

    def fake_launch(argv, n, **kw):
        called["argv"], called["n"] = argv, n
        return 0
    monkeypatch.setattr(bench, "launch_ranks", fake_launch)
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == 0 and called["n"] == 4
    assert called["argv"][0].endswith("bench.py") and called["argv"][1:] == ["--gpus", "4", "--steps", "2"]


def test_mfc_env_is_recorded(monkeypatch):
    monkeypatch.setenv("MFC_CNX_MAX_BLOCKS", "1024")
    monkeypatch.setenv("NOT_MFC", "1")
    e = bench.mfc_env()
    assert e.get("MFC_CNX_MAX_BLOCKS") == "1024" and "NOT_MFC" not in e


# ---------------------------------------------------------------------------------------------------------------
# precision of the gradient exchange: bf16 vs fp32 summation across ranks (distributed.py, "Precision of the exchange")
# ---------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _sum_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(7 + rank)
        n = 1 << 20
        # per-rank gradient contributions of one big kernel: common signal + rank noise, as the shards of a batch give
        sig = torch.randn(n, generator=torch.Generator().manual_seed(99))
        mine = (sig + 0.5 * torch.randn(n, generator=g)) * 1e-3
        b16 = mine.bfloat16()                          # what each rank's weight-gradient GEMM stores
        ref = b16.double()
        dist.all_reduce(ref)                            # exact sum of the bf16 contributions
        s16 = b16.clone()
        dist.all_reduce(s16)                            # the exchange as shipped: bf16 on the wire and in the sum
        s32 = b16.float()
        dist.all_reduce(s32)
        s32 = s32.bfloat16()                            # fp32 exchange, rounded once for the bf16 gradient buffer
        den = b16.abs().double()
        dist.all_reduce(den)                            # sum over ranks of |contribution|: bounds every partial sum
        den = den.clamp_min(1e-30)
        e16 = ((s16.double() - ref).abs() / den).max().item()
        e32 = ((s32.double() - ref).abs() / den).max().item()
        r16 = ((s16.double() - ref).norm() / ref.norm()).item()
        r32 = ((s32.double() - ref).norm() / ref.norm()).item()
        q.put((rank, e16, e32, r16, r32))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_bf16_exchange_error_bound(world):
    """The big kernels' gradients are summed across ranks in bf16.  With u = 2^-8 (half a bf16 ulp, relative) and
    A = sum over ranks of |contribution| (which bounds every partial sum): an fp32 exchange rounded once into the bf16
    gradient buffer is within u A of the exact sum; the bf16 exchange rounds the running sum after each of the
    world - 1 additions a ring (gloo here, RCCL on the GPUs) or a tree makes on an element's path, so it is within
    (world - 1) u A in the worst case, and its RMS error stays within sqrt(world - 1) of the single rounding -- the
    bound DESIGN.md section 0(e) states.  (The AdamW update divides the gradient by its running RMS, so an error of
    this size moves an update by about that fraction of lr.)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sum_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(len(r) == 5 for r in res), res
    u = 2.0 ** -8
    import math
    for _, e16, e32, r16, r32 in res:
        assert e32 <= u * 1.001                              # one rounding
        assert e16 <= u * (world - 1) * 1.001                # one rounding per addition on the element's path
        assert r32 < 0.6 * u and r16 < 0.6 * u * math.sqrt(world - 1)
        assert r16 >= r32 * 0.99                             # the bf16 exchange is never the more accurate one
