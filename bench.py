#!/usr/bin/env python3
"""bench.py -- throughput of the global-alignment hot path on MI355X (BASELINE.json metric: GCUPS +
aligned pairs/s on the 250 bp x 1 M-seq DB shape at --id 0.9).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" = one pass of the hot path over one batch: every (query, candidate) pair of 100 k x 250 bp
queries against their 8 candidates in a device-resident 1 M x 1 kbp family-structured DB
(BASELINE config[1]) -> packed-int16 DP kernel (scores + direction bits) + traceback kernel
(statistics + CIGAR run lists).  Inputs (DB, queries, pair/task list) are resident in HBM before the
timed region; results stay in HBM (the PCIe fetch is timed separately and reported, never in `value`).

Rank 0 prints ONE JSON line (see DESIGN.md for `roofline` / `cpu_baseline`).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

# peak of the bounding unit (packed int16 VALU):  256 CU x 4 SIMD x 16 lanes/clk x 2 (v_pk_*_i16) x 2.4 GHz
# (MI355X_MICROARCH.md: 256 CUs, 4 SIMDs/CU, 2400 MHz; VOP3P issues at 16 lanes/clk/SIMD -- measured with
#  vsearch_amd/csrc/ubench_valu.hip: 69-73 T int16-ops/s, profiles/r01_ubench_valu.txt)
PEAK_INT16_TOPS = 256 * 4 * 16 * 2 * 2.4e9 / 1e12
# ops per DP cell.  SURVEY.md 8(d) prescribes 15 = the reference's onestep (align_simd.cpp:765-780: add + 4 sub + 4 max + min +
# max + 4 direction compares).  The checkpointing DP kernel does NOT execute 15: its general row body is score pack + add +
# 2 max + 4 sub + 2 max = 10 packed-int16 instructions per cell pair; the 4 direction compares run only on the tiles the
# traceback crosses, the min/max only for tasks that can overflow.  Since r01k the bench workload runs in TILTED coordinates
# (DESIGN.md 4.2): the interior row body is perm + add + 2 max + sub + 2 max = 7 instructions, two of them (add, sub) 32-bit
# ops at twice the VOP3P rate = 6.0 VOP3P issue slots; row R-1 of a lane needs 9.5.  `roofline.frac` prices the kernel with
# the issue slots its own row body needs (the honest "how much of the VALU is doing recurrence work" number); the 10-op and the
# SURVEY 15-op accountings are carried next to it (both exceed or approach 1 by construction: that work is no longer executed).
SURVEY_OPS_PER_CELL = 15
GENERAL_OPS_PER_CELL = 10


def ops_per_cell(info):
    """VOP3P issue slots per DP cell of the dominant launch's row body (see the comment above); info = Plan.describe()"""
    rows = max(1, info["rows_dominant"])
    if 2 * info["tasks_tilted"] < info["tasks"]:
        return float(GENERAL_OPS_PER_CELL)
    return ((rows - 1) * 6.0 + 9.5) / rows


def traffic_from_profile(a, world):
    """HBM bytes per DP-kernel launch from the committed PMC passes (profiles/r01_traffic.json, written by
    profiles/summarize.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of THIS command); counters cannot be
    collected from inside the timed run, so the figure is only reported for the workload it was measured on."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as f:
            t = json.load(f)
        w = t["workload"]
        if world == 1 and all(getattr(a, k) == w[k] for k in ("queries", "qlen", "db", "dlen", "cands")):
            return t["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--queries", type=int, default=100_000)
    ap.add_argument("--qlen", type=int, default=250)
    ap.add_argument("--db", type=int, default=1_000_000)
    ap.add_argument("--dlen", type=int, default=1000)
    ap.add_argument("--cands", type=int, default=8)
    ap.add_argument("--cpu-pairs", type=int, default=400_000, help="pairs in the bounded CPU-baseline sample")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = all online cores")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--dir-budget-gb", type=float, default=0.0)
    return ap.parse_args()


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)     # RCCL

    from vsearch_amd import Aligner, SequenceSet, workload

    # ---- synthetic inputs, generated in HBM (weak scaling: every rank holds a DB replica and its own queries)
    t0 = time.time()
    db_ascii, db_off, db_len, fam = workload.make_family_db(a.db, a.dlen, seed=17, device=dev)
    q_ascii, q_off, q_len, src = workload.make_queries(db_ascii, db_off, db_len, a.queries, a.qlen,
                                                       seed=11 + 1000 * rank, device=dev)
    qidx, tidx = workload.family_candidates(src, fam, per_query=a.cands, seed=5 + rank)
    torch.cuda.synchronize()
    t_gen = time.time() - t0

    al = Aligner(device=local_rank)
    T = SequenceSet(al, blob=db_ascii.numel(), offsets=db_off, lengths=db_len, device_ptr=db_ascii.data_ptr())
    Q = SequenceSet(al, blob=q_ascii.numel(), offsets=q_off, lengths=q_len, device_ptr=q_ascii.data_ptr())
    t0 = time.time()
    plan = al.plan(Q, T, qidx, tidx, dir_budget_bytes=int(a.dir_budget_gb * (1 << 30)))
    t_plan = time.time() - t0             # host: pairs -> wavefront tasks, task upload, checkpoint buffer (one-time hipMalloc)
    n_pairs = len(qidx)
    cells = int((q_len[qidx].astype(np.int64) * db_len[tidx].astype(np.int64)).sum())
    hits = torch.empty(n_pairs * 24, dtype=torch.uint8, device=dev)
    gathered = torch.empty(world * n_pairs * 24, dtype=torch.uint8, device=dev) if world > 1 else None

    def step():
        plan.run()
        tm = plan.sync()
        if world > 1:
            # the only collective on the path: final gather of the fixed-size hit records over RCCL/xGMI
            plan.export_hits(hits.data_ptr(), hits.numel())
            dist.all_gather_into_tensor(gathered, hits)
        return tm

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    fwd_ms = tb_ms = 0.0
    fwd_launches = 0
    tm = None
    for _ in range(a.steps):
        tm = step()
        fwd_ms += tm.forward_ms
        tb_ms += tm.traceback_ms
        fwd_launches += tm.forward_launches
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- PCIe-inclusive rate (results + CIGAR text on the host), informational ----
    t0 = time.perf_counter()
    res = plan.fetch()
    t_fetch = time.perf_counter() - t0

    total_cells = cells * world * a.steps
    value = total_cells / elapsed / 1e9
    ms_per_step = elapsed / a.steps * 1e3
    fwd_avg_ms = fwd_ms / max(1, fwd_launches)
    cells_per_launch = cells * a.steps / max(1, fwd_launches)
    info = plan.describe()
    tilted = 2 * info["tasks_tilted"] >= info["tasks"]
    OPS_PER_CELL = ops_per_cell(info)
    achieved = cells_per_launch * OPS_PER_CELL / (fwd_avg_ms * 1e-3) / 1e12
    out = {
        "metric": "GCUPS (useful DP cells/s of the search16 global-alignment path: DP + traceback + CIGAR)",
        "value": round(value, 2),
        "unit": "GCUPS",
        "pairs_per_s": round(n_pairs * world * a.steps / elapsed, 1),
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int16", "data": "synthetic",
        "config": {
            "workload": f"usearch_global candidate batch: {a.queries} x {a.qlen} bp queries vs {a.db} x {a.dlen} bp "
                        f"family-structured DB (device-resident), {a.cands} candidates/query = {n_pairs} pairs/GPU/step, "
                        "--id 0.9 shape, default scoring (BASELINE config[1])",
            "candidates": "synthetic: source member + 7 same-family members (k-mer heuristic is host-side, outside the path)",
            "parallelism": f"query-sharded x{world}, DB replicated" if world > 1 else "single GPU",
            "cells_per_step_per_gpu": cells,
        },
        "roofline": {
            "kernel": f"vsx_forward_kernel<{info['rows_dominant']},true,false,true,true>" if tilted else f"vsx_forward_kernel<{info['rows_dominant']},true,...>",
            "bound": "valu-int16",
            "achieved": round(achieved, 3),
            "peak": round(PEAK_INT16_TOPS, 2),
            "unit": "Tops/s",
            "frac": round(achieved / PEAK_INT16_TOPS, 4),
            "ops_per_cell": round(OPS_PER_CELL, 3),
            "coordinates": "tilted" if tilted else "plain",
            "plan": info,
            "general_row_body_accounting": {"ops_per_cell": GENERAL_OPS_PER_CELL,
                                            "frac": round(achieved * GENERAL_OPS_PER_CELL / OPS_PER_CELL / PEAK_INT16_TOPS, 4)},
            "survey_accounting": {"ops_per_cell": SURVEY_OPS_PER_CELL,
                                  "achieved": round(achieved * SURVEY_OPS_PER_CELL / OPS_PER_CELL, 3),
                                  "frac": round(achieved * SURVEY_OPS_PER_CELL / OPS_PER_CELL / PEAK_INT16_TOPS, 4),
                                  "note": "SURVEY 8(d) counts the reference's onestep incl. 4 direction compares + min/max; the kernel "
                                          "does not execute those per cell, so this exceeds 1 by construction"},
            "kernel_ms_avg": round(fwd_avg_ms, 3),
            "kernel_launches": fwd_launches,
            "kernel_gcups": round(cells_per_launch / (fwd_avg_ms * 1e-3) / 1e9, 1),
            "traffic": traffic_from_profile(a, world),
            "hbm_algorithmic_bytes_per_launch": int(tm.dir_bytes / max(1, tm.forward_launches)),
            "hbm_algorithmic_GBps": round(tm.dir_bytes / max(1, tm.forward_launches) / (fwd_avg_ms * 1e-3) / 1e9, 1),
        },
        "kernel_split_ms_per_step": {"forward": round(fwd_ms / a.steps, 3), "traceback": round(tb_ms / a.steps, 3)},
        "plan_s": round(t_plan, 3),
        "fetch_s": round(t_fetch, 3),
        "value_incl_fetch": round(cells / (ms_per_step * 1e-3 + t_fetch) / 1e9, 2),
        "gen_s": round(t_gen, 2),
    }

    # ---- CPU baseline: the reference's own SSE2 search16 (oracle/_ref, built from /root/reference) on a
    #      bounded sample of the SAME workload; falls back to the scalar port if _ref was not shipped ----
    if not a.no_cpu:
        try:
            out["cpu_baseline"] = cpu_baseline(a, db_ascii, db_off, db_len, q_ascii, q_off, q_len, qidx, tidx, res)
        except Exception as e:  # the baseline is a side measurement: never lose the bench line over it
            out["cpu_baseline"] = {"error": repr(e)}
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def usable_cpus():
    """online CPUs this process may actually use: affinity mask, capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(a, db_ascii, db_off, db_len, q_ascii, q_off, q_len, qidx, tidx, res):
    from oracle import pyoracle
    n = min(a.cpu_pairs, len(qidx))
    n -= n % a.cands
    qi, ti = qidx[:n], tidx[:n]
    uq, qinv = np.unique(qi, return_inverse=True)
    ut, tinv = np.unique(ti, return_inverse=True)

    def gather(flat, off, ln, ids):
        offs = torch.from_numpy(off[ids].astype(np.int64)).to(flat.device)
        lens = ln[ids].astype(np.int64)
        L = int(lens.max())
        idx = offs[:, None] + torch.arange(L, device=flat.device)[None, :]
        idx = torch.clamp(idx, max=flat.numel() - 1)
        m = flat[idx].cpu().numpy()
        blob = b"".join(m[k, :lens[k]].tobytes() for k in range(len(ids)))
        o = np.zeros(len(ids), np.uint64)
        o[1:] = np.cumsum(lens[:-1])
        return blob, o, lens.astype(np.uint32)

    qb, qo, ql = gather(q_ascii, q_off, q_len, uq)
    tb, to, tl = gather(db_ascii, db_off, db_len, ut)
    sample_cells = int((ql[qinv].astype(np.int64) * tl[tinv].astype(np.int64)).sum())
    threads = a.cpu_threads or usable_cpus()
    if pyoracle.have_ref():
        ref = pyoracle.Reference()
        # groups: consecutive pairs of one query (the reference's search16 call shape)
        gq = qinv[::a.cands].astype(np.uint32)
        goff = np.arange(0, n + 1, a.cands, dtype=np.uint64)
        one_s, _, chk1 = ref.time_groups(qb, qo, ql, tb, to, tl, gq[:max(1, len(gq) // 16)],
                                         goff[:max(1, len(gq) // 16) + 1], tinv.astype(np.uint32), threads=1)
        one_cells = int((ql[qinv[:int(goff[max(1, len(gq) // 16)])]].astype(np.int64) *
                         tl[tinv[:int(goff[max(1, len(gq) // 16)])]].astype(np.int64)).sum())
        secs, c, chk = ref.time_groups(qb, qo, ql, tb, to, tl, gq, goff, tinv.astype(np.uint32), threads=threads)
        gpu_chk = int(res.score[:n].astype(np.int64).sum() + res.aligned[:n].astype(np.int64).sum()
                      + res.matches[:n].astype(np.int64).sum())
        return {"value": round(c / secs / 1e9, 2), "unit": "GCUPS", "cores": threads, "kind": "reference",
                "sample": f"reference SSE2 search16 (oracle/_ref, -O3 -march=x86-64) on the first {n} pairs "
                          f"({len(uq)} queries x {a.cands}) of the same step, {threads} std::threads "
                          f"(= usable host CPUs: affinity {len(os.sched_getaffinity(0))}, cgroup quota applied; "
                          f"{os.cpu_count()} online), {secs:.2f} s wall",
                "single_thread_GCUPS": round(one_cells / one_s / 1e9, 2),
                "parity_checksum_match": bool(chk == gpu_chk)}
    orc = pyoracle.Oracle()
    m = min(n, 2000)
    t0 = time.perf_counter()
    sc, al, ma, mm, g, cig = orc.align_batch(qb, qo, ql, tb, to, tl, qinv[:m], tinv[:m])
    secs = time.perf_counter() - t0
    c = int((ql[qinv[:m]].astype(np.int64) * tl[tinv[:m]].astype(np.int64)).sum())
    ok = bool(np.array_equal(sc, res.score[:m]) and cig == res.cigar[:m])
    return {"value": round(c / secs / 1e9, 3), "unit": "GCUPS", "cores": 1, "kind": "port",
            "sample": f"scalar oracle (oracle/nw_oracle.c) on the first {m} pairs of the same step, {secs:.2f} s",
            "parity_match": ok}


if __name__ == "__main__":
    main()
