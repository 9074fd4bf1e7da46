"""world_size-2 gloo tests (CPU) of the multi-GPU plumbing used by bench.py --gpus N."""
import os
import socket
import subprocess
import sys
import textwrap

from sunscreen_amd.dist import shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_the_batch():
    for total in (0, 1, 7, 8, 1024, 4097):
        for world in (1, 2, 3, 8):
            blocks = [shard_range(total, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            for (a, b), (c, d) in zip(blocks, blocks[1:]):
                assert b == c
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo_timing_and_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(
        textwrap.dedent(
            """
            import sys, time
            sys.path.insert(0, %r)
            import torch, torch.distributed as dist
            from sunscreen_amd import dist as D
            rank, local_rank, world = D.init("gloo")
            assert world == 2
            total = 9
            lo, hi = D.shard_range(total, rank, world)
            # each rank "processes" its block: result item i = i * 10 + per-item payload
            local = torch.stack([torch.full((2, 3), i * 10, dtype=torch.int64) for i in range(lo, hi)])
            calls = []
            def step():
                calls.append(1)
                time.sleep(0.01 * (rank + 1))   # rank 1 is slower: the reported time must be the max
            elapsed = D.timed_steps(step, steps=3, warmup=1)
            assert len(calls) == 4
            assert elapsed >= 0.06 - 1e-3, elapsed
            # one-time key replication: the owner's serialised bytes reach every rank unchanged
            blob = bytes(range(256)) * 1000 + b"tail"
            got = D.broadcast_bytes(blob if rank == 1 else None, src=1)
            assert got == blob
            full = D.gather_results(local, total, root=1)   # gather to ONE root (SURVEY 8e), not an all_gather
            if rank == 1:
                assert full.shape == (total, 2, 3)
                assert [int(full[i, 0, 0]) for i in range(total)] == [i * 10 for i in range(total)]
            else:
                assert full is None
            # SURVEY 8e "Exception" (examples/pir, database sharded by row): one partial ciphertext per rank, summed on the
            # root with the evaluator's modular addition (here: a stand-in add modulo one prime per residue row)
            q = torch.tensor([97, 193], dtype=torch.int64).view(1, 1, 2, 1)
            g = torch.Generator().manual_seed(7)
            parts = [torch.randint(0, 97, (1, 2, 2, 8), generator=g, dtype=torch.int64) for _ in range(world)]
            calls2 = []
            def add(a, b):
                calls2.append(1)
                return (a + b) %% q
            tot = D.reduce_ciphertexts(parts[rank], add, root=0)
            if rank == 0:
                assert torch.equal(tot, (parts[0] + parts[1]) %% q) and len(calls2) == world - 1
            else:
                assert tot is None and not calls2
            # the client's query reaches every shard: one tensor from the root, in messages below 1 GiB
            qt = torch.arange(24, dtype=torch.int64).view(2, 3, 4) if rank == 0 else None
            got_q = D.broadcast_tensor(qt, (2, 3, 4), torch.int64, "cpu", src=0)
            assert torch.equal(got_q, torch.arange(24, dtype=torch.int64).view(2, 3, 4))
            D.barrier()
            D.barrier()
            dist.destroy_process_group()
            print("rank", rank, "ok")
            """
        )
        % ROOT
    )
    port = _free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
    procs = [
        subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        for r in range(2)
    ]
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"rank {r} ok" in o


def test_a_single_rank_group_issues_its_collectives_when_forced(tmp_path):
    """HIPBFV_DIST_FORCE=1 (how tests/test_gpu_dist.py runs the RCCL path on a one-GPU box): a process group of one rank is not
    short-circuited -- the collectives are issued (here over gloo) and return the payloads unchanged; without the variable the
    module does not touch torch.distributed at world_size 1."""
    script = tmp_path / "solo.py"
    script.write_text(
        textwrap.dedent(
            """
            import os, sys
            sys.path.insert(0, %r)
            import torch, torch.distributed as dist
            from sunscreen_amd import dist as D
            assert D.solo()                      # no process group yet
            rank, local_rank, world = D.init("gloo")
            assert world == 1 and dist.is_initialized() and not D.solo() and not D.is_nccl()
            calls = {"n": 0}
            real = dist.broadcast
            def counting(*a, **k):
                calls["n"] += 1
                return real(*a, **k)
            dist.broadcast = counting
            blob = bytes(range(200)) * 50
            assert D.broadcast_bytes(blob, 0) == blob and calls["n"] == 2      # length + one payload message
            t = torch.arange(12, dtype=torch.int64).reshape(3, 2, 2)
            assert torch.equal(D.broadcast_tensor(t, t.shape, t.dtype, "cpu", 0), t) and calls["n"] == 3
            assert torch.equal(D.gather_results(t, 3), t)
            assert torch.equal(D.reduce_ciphertexts(t, lambda a, b: a + b), t)
            D.barrier()
            assert D.timed_steps(lambda: None, 2, 1) >= 0.0
            os.environ.pop("HIPBFV_DIST_FORCE")
            assert D.solo()                      # the default: a world of one needs no collective
            dist.destroy_process_group()
            print("SOLO_OK")
            """
            % ROOT
        )
    )
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", HIPBFV_DIST_FORCE="1")
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "SOLO_OK" in p.stdout, p.stderr[-2000:]
