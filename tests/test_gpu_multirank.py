"""-m gpu: the multi-GPU path with world_size 2 on ONE GPU: two ranks, each with its own context on device 0,
collectives over gloo.  The code under test is what bench.py --gpus N runs per rank (vsearch_amd/sharding.py):
query-sharded alignment + the final gather of hit records AND CIGAR run words; row-sharded allpairs."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, random
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from vsearch_amd import Aligner, SearchSession, sharding
from tests import common
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
rng = random.Random(77)
db, fam = common.family_db(rng, 8, 8, 320, div=0.07)
db += ["", "ACGTNNRYacgtu" * 9]
qs, src = common.queries_from_db(rng, db[:64], 41, 150)
qs += ["", common.rnd_seq(rng, 200, "ACGTRYN")]
qidx = np.repeat(np.arange(len(qs)), 5)
tidx = np.array([rng.randrange(len(db)) for _ in qidx])
with Aligner(device=0) as al:
    Q, T = al.sequences(qs), al.sequences(db)
    rec_all, runs_all, counts, order = sharding.sharded_align(al, Q, T, qidx, tidx, len(qs), dist)
    assert sum(counts) == len(qidx) and sorted(order.tolist()) == list(range(len(qidx)))
    assert counts[rank] == int(((qidx >= sharding.shard_queries(len(qs), world, rank)[0]) & (qidx < sharding.shard_queries(len(qs), world, rank)[1])).sum())
    # every rank holds everything: compare with a single-rank run of ALL pairs on this rank's GPU context
    one = al.align_pairs(Q, T, qidx, tidx)
    d = sharding.decode_records(rec_all)
    cig = sharding.cigars_from_gather(rec_all, runs_all)
    for j, g in enumerate(order):
        got = (int(d["score"][j]), int(d["aligned"][j]), int(d["matches"][j]), int(d["mismatches"][j]), int(d["gaps"][j]), cig[j])
        assert got == one.row(int(g)), (rank, j, int(g), got, one.row(int(g)))
    assert sum(1 for c in cig if c) > 150

    # allpairs: interleaved rows (config 4's sharding); the union over the ranks must be the single-rank result
    ss = SearchSession(al, db[:48], id=0.8)
    rows = sharding.shard_allpairs_rows(48, world, rank)
    mine = ss.allpairs_rows(rows)
    allh = [None] * world
    dist.all_gather_object(allh, (rows.tolist(), mine))
    full = ss.allpairs(0, 48)
    merged = [None] * 48
    for rws, hl in allh:
        for r, h in zip(rws, hl):
            assert merged[r] is None
            merged[r] = h
    assert merged == full
    assert sum(len(h) for h in full) > 100

    # --usearch_global sharded over the ranks (config 5's form): block of queries per rank, one gather of hit structs +
    # counts + CIGAR text to rank 0 == the single-rank search of all queries, hit for hit, field for field
    for kw in (dict(id=0.8, maxaccepts=3, maxrejects=8), dict(id=0.8, maxaccepts=2, strand_both=1, soft_mask=2)):
        ss2 = SearchSession(al, db, **kw)
        got = sharding.sharded_search(ss2, qs, dist, dst=0)
        if rank == 0:
            exp = ss2.search_batch(qs)
            assert SearchSession.hits_as_lists(*got) == exp
            assert sum(len(h) for h in exp) > 40
        else:
            assert got is None
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_sharded_alignment_and_gather_world2(gpu_required, tmp_path):
    script = tmp_path / "worker_gpu.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29547", str(script), ROOT]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert p.stdout.count("ok") == 2


def _same_hits(a, b):
    """(first, hits, cigar blob) of two runs: same lists, every field; CIGARs compared as text (blob offsets may differ)"""
    import numpy as np
    fa, ha, ca = a
    fb, hb, cb = b
    assert np.array_equal(fa, fb)
    assert len(ha) == len(hb)
    for name in ha.dtype.names:
        if name in ("cigar_off", "pad"):
            continue
        assert np.array_equal(ha[name], hb[name]), name
    for x in range(len(ha)):
        oa, ob = int(ha["cigar_off"][x]), int(hb["cigar_off"][x])
        assert ca[oa:ca.index(b"\0", oa)] == cb[ob:cb.index(b"\0", ob)], x


@pytest.mark.parametrize("devices", [(0, 0), (0, 0, 0)])
def test_multi_device_handle_equals_single_device(gpu_required, devices):
    """vsx_multi_searcher (vsx_multi.cpp): one C handle over several device replicas -- here two / three replicas on ONE GPU -- must
    return exactly what a single searcher returns: --usearch_global (contiguous query blocks merged in query order; plus strand
    and both strands + DUST + abundances / labels) and --allpairs_global (rows dealt boustrophedon, merged per row).
    VERDICT r02 'next' #5: the multi-device driver below the C-ABI."""
    import random
    from tests import common
    from vsearch_amd import Aligner
    from vsearch_amd.search import SearchSession, MultiSearchSession
    rng = random.Random(303)
    db, fam = common.family_db(rng, 10, 8, 330, div=0.06)
    db += ["", "ACGTNNRYacgtu" * 9]
    qs, src = common.queries_from_db(rng, db[:80], 53, 160)
    qs += ["", common.rnd_seq(rng, 210, "ACGTRYN"), db[3]]
    qsize = [rng.randint(1, 40) for _ in qs]
    qlab = [f"q{i};size={s}" for i, s in enumerate(qsize)]
    dsize = [rng.randint(1, 40) for _ in db]
    dlab = [f"t{i}" for i in range(len(db))]
    for kw in (dict(id=0.8, maxaccepts=3, maxrejects=8),
               dict(id=0.8, maxaccepts=2, strand_both=1, soft_mask=2, minsizeratio=0.2, self_=1)):
        with Aligner(device=0) as al:
            one = SearchSession(al, db, sizes=dsize, labels=dlab, **kw)
            exp = one.search_batch_raw(qs, sizes=qsize, labels=qlab)
            exp_pairs = one.stats["pairs_aligned"]
            exp_ap = None
            if not kw.get("strand_both"):
                f, h, c = [], [], b""
                lib_hits = one.allpairs(0, 60)
            one.close()
        with MultiSearchSession(db, devices=devices, sizes=dsize, labels=dlab, **kw) as ms:
            assert ms.n_devices == len(devices)
            got = ms.search_batch_raw(qs, sizes=qsize, labels=qlab)
            assert ms.stats["pairs_aligned"] == exp_pairs
            _same_hits(got, exp)
            assert len(got[1]) > 40
            if not kw.get("strand_both"):
                f, h, c = ms.allpairs_raw(0, 60)
                got_lists = SearchSession.hits_as_lists(f, h, c)
                assert got_lists == lib_hits
                assert sum(len(x) for x in got_lists) > 100


@pytest.mark.parametrize("mode,gather", [("strong", "async"), ("weak", "async"), ("strong", "sync")])
def test_bench_two_ranks_on_one_gpu(gpu_required, mode, gather):
    """bench.py --gpus 2 exactly as the driver launches it (torch.distributed.run, one rank per 'GPU' -- here both on device 0, gloo
    instead of RCCL): --scaling strong cuts ONE job of --queries queries into two blocks, weak gives every rank its own; the final
    gather is the asynchronous fixed-capacity collective (sharding.FixedGather) or the synchronous one.  Rank 0's line must carry the
    whole job's pairs and a passing gather_check (its own block back unchanged, the other rank's run offsets rebased past it)."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    nq = 6000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29571", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--backend", "gloo",
           "--scaling", mode, "--gather", gather, "--queries", str(nq), "--db", "20000", "--dlen", "600", "--qlen", "200"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, p.stdout[-2000:]
    d = json.loads(line[0])
    assert d["n_gpus"] == 2 and d["scaling"] == mode and d["gather_check"] is True
    pairs_per_step = d["pairs_per_s"] * d["ms_per_step"] * 1e-3
    expect = nq * 8 * (1 if mode == "strong" else 2)
    assert abs(pairs_per_step - expect) < 0.01 * expect, (pairs_per_step, expect)


@pytest.mark.parametrize("gather", ["async", "sync"])
def test_bench_single_rank_over_rccl(gpu_required, gather):
    """r06: the N > 1 code path of bench.py with ONE rank and the `nccl` backend -- RCCL itself executes every collective of the path on the
    one GPU of this box (it refuses two ranks on one device: profiles/r06/r06z_rccl_two_ranks_one_gpu.txt): process-group init with a
    device id, the rank-identity all-gather, vsx_plan_export_hits / _runs into device tensors, the asynchronous fixed-capacity gather
    (non-blocking gather + flag all-reduce, Work.is_completed / wait) or the synchronous all-gathers, the timing all-reduces, gather_check.
    The collectives are degenerate; the tensors, dtypes, devices and call sequence are the ones an 8-GPU run issues."""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29581", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    nq = 6000
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--single-rank-dist", "--backend", "nccl", "--gather", gather, "--kernels-only",
           "--steps", "4", "--warmup", "1", "--queries", str(nq), "--db", "20000", "--dlen", "600", "--qlen", "200"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["gather_check"] is True, d.get("gather_check")
    r = d["rccl"]
    assert r["backend"] == "nccl" and r["world"] == 1 and r["nccl_version"] and r["device_of_rank"] == [0]
    if gather == "async":
        st = r["fixed_gather_overlap"]
        assert st["collects"] + (1 if st["degraded_to_sync"] else 0) >= 4 and r["fixed_gather_sync_steps"] == 0, r
    pairs_per_step = d["pairs_per_s"] * d["ms_per_step"] * 1e-3
    assert abs(pairs_per_step - nq * 8) < 0.01 * nq * 8


def test_dry_collectives_single_rank_over_rccl(gpu_required):
    """bench.py --dry-collectives on one rank with the nccl backend: the gather path alone on DEVICE tensors through RCCL -- synchronous gather,
    the asynchronous fixed-capacity form incl. overflow steps (the synchronous redo inside collect), the collective degrade vote"""
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29582", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--single-rank-dist", "--backend", "nccl", "--dry-collectives"],
                       env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["dry_collectives"] == {"gather_results_to_rank0": True, "fixed_gather_async_incl_one_rank_overflow": True, "fixed_gather_sync_steps": 1,
                                     "fixed_gather_overlap_samples": True, "fixed_gather_mixed_votes_one_rank_overflows": True,
                                     "fixed_gather_degraded_at_step": 5}, d
    assert d["rccl"]["backend"] == "nccl" and d["rccl"]["nccl_version"]
