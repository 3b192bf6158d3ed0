#!/usr/bin/env python3
"""bench.py -- throughput of the global-alignment hot path on MI355X (BASELINE.json metric: GCUPS +
aligned pairs/s on the 250 bp x 1 M-seq DB shape at --id 0.9).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" = one pass of the hot path over one batch: every (query, candidate) pair of 100 k x 250 bp
queries against their 8 candidates in a device-resident 1 M x 1 kbp family-structured DB
(BASELINE config[1]) -> packed-int16 DP kernel (scores + checkpoints) + traceback kernel (statistics +
CIGAR run lists) + text kernel (CIGAR strings, output arrays).  Inputs (DB, queries, pair/task list) are
resident in HBM before the timed region; results stay in HBM -- that is `value`.  Next to it, never in it:
  end_to_end        vsx_align_pairs on the same pair list, host arrays in / host arrays + CIGAR text out
                    (planning, kernels, PCIe, pipelined slices; pools warm), steady state
  search_end_to_end vsx_search_batch (= --usearch_global --id 0.9) for the same queries against the same DB:
                    k-mer candidate stage + dispatch + alignments + hit lists, next to the reference CLI
                    (oracle/_ref/vsearch_ref, all usable cores) on a sample of the same queries
With --gpus N every rank plays one block of a query-sharded job (vsearch_amd/sharding.py): own queries, DB
replica, no data-path collective, one gather of hit records + CIGAR run words at the end of every step.

Rank 0 prints ONE JSON line (see DESIGN.md for `roofline` / `cpu_baseline`).
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import threading
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

# ---- roofline of the DP kernel: VALU ISSUE.  r03 calibration (profiles/r03/r03_ubench_rowbody.txt, vsearch_amd/csrc/ubench_valu.hip
# UB_ROWBODY / UB_MAX3): in the kernel's instruction mix EVERY VALU instruction -- the 32-bit add / subtract as much as the packed
# maxima and v_perm_b32 -- costs one ~4-cycle issue slot per wave (4.1-4.2 measured; the 2-cycle VOP2 rate of r01's per-class
# micro-benchmark only shows in pure VOP2 streams of two or more waves).  So the honest unit is INSTRUCTIONS:
#   peak    = 256 CU x 4 SIMD x 2.4 GHz / 4 cycles = 6.14e11 wave-instructions/s  (x 64 lanes = 39.3 T lane-instructions/s)
#   useful  = the recurrence's row body, per lane-row (= 2 cells: one row of two targets):
#             MAX3 class (r03): v_perm_b32, v_add_u32, v_pk_maximum3_f16, v_sub_u32, 2 x v_pk_max_u16 = 6; row R-1 of a lane = 9
#             TILT class (r01k): 7 (two v_pk_max_u16 for H); row R-1 = 9.5;  general row body = 10
#   frac    = useful instructions issued per second / peak = the share of ALL VALU issue slots of the launch that the recurrence
#             itself needs; the rest is per-step work (DPP hand-over, LDS addressing, checkpoint packing, last-row tracking),
#             pipeline fill and drain, and idle issue.
# The measured ceiling of profiles/r01_ubench_valu.txt (69-73 T int16-op/s = 0.88-0.93 of the nominal 78.6) applies to the issue rate
# as well: `frac_of_measured_ceiling` divides by 0.90 of the nominal peak.
# SURVEY.md 8(d)'s 15 ops/cell (the reference's onestep incl. 4 direction compares + min/max per cell) is NOT a bound for this
# algorithm -- the compares run only on the tiles the traceback crosses, the tilt removes two subtractions: at 15 ops/cell the kernel
# would "exceed" the peak (1.5) -- and is therefore not part of the headline block any more (DESIGN.md 4.1 keeps the derivation).
PEAK_WAVE_INSTR_PER_S = 256 * 4 * 2.4e9 / 4.0
PEAK_LANE_TOPS = PEAK_WAVE_INSTR_PER_S * 64 / 1e12
MEASURED_CEILING = 0.90
CPI_CALIBRATED = 4.1            # issue cycles per executed VALU instruction of the steady loop's mix (profiles/r03/r03_ubench_rowbody.txt)


def row_body_instructions(info):
    """VALU instructions per lane-row (2 cells) of the dominant launch's row body; info = Plan.describe()"""
    rows = max(1, info["rows_dominant"])
    if 2 * info["tasks_tilted"] < info["tasks"]:
        return 10.0
    if 2 * info.get("tasks_max3", 0) >= info["tasks"]:
        return ((rows - 1) * 6.0 + 9.0) / rows
    return ((rows - 1) * 7.0 + 9.5) / rows


def sparse_nq(info):
    """tasks per wave of the dominant launch as vsx_plan_describe lets it be seen: 1 unless most tasks sit in a sparse-task class"""
    if 2 * info.get("tasks_sparse", 0) < info["tasks"]:
        return 1
    w = max(1, info.get("waves", info["tasks"]))
    return 4 if info["tasks"] / w > 3.0 else 2


KERNEL_SOURCES = ("vsearch_amd/csrc/vsx_device.hip", "vsearch_amd/csrc/vsx_internal.h", "vsearch_amd/csrc/vsx_tbtext.hip")


def kernel_source_sha():
    """identity of the kernels a PMC file was measured on: sha256 over the kernel sources (first 16 hex digits)"""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def counters_from_profile(a, world):
    """PMC results of THIS command on THESE kernels (profiles/pmc_current.json, written by profiles/run_profile.sh ->
    summarize.py from separate rocprofv3 --pmc passes; counters cannot be collected from inside the timed run).  The file
    carries the sha of the kernel sources it was measured on: a stale file (kernels changed since) or another workload is
    refused and `traffic` is null."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_current.json")) as f:
            t = json.load(f)
    except Exception:
        return None, "no profiles/pmc_current.json"
    if t.get("kernel_source_sha") != kernel_source_sha():
        return None, f"profiles/pmc_current.json was measured on kernel sources {t.get('kernel_source_sha')}, HEAD has {kernel_source_sha()}: refused"
    w = t.get("workload", {})
    if world != 1 or any(getattr(a, k) != w.get(k) for k in ("queries", "qlen", "db", "dlen", "cands")):
        return None, "profiles/pmc_current.json was measured on another workload"
    return t, None


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
    ap.add_argument("--no-cpu", action="store_true", help="skip every CPU leg (cpu_baseline and the reference CLI)")
    ap.add_argument("--kernels-only", action="store_true", help="only the timed kernel steps (profiling passes)")
    ap.add_argument("--e2e-calls", type=int, default=5, help="timed vsx_align_pairs calls of the end-to-end figure")
    ap.add_argument("--no-search", action="store_true", help="skip the vsx_search_batch end-to-end figure")
    ap.add_argument("--no-shapes", action="store_true", help="skip the kernels-only runs of the other BASELINE shapes (configs 3 / 4 / 5)")
    ap.add_argument("--search-mask", choices=["none", "dust"], default="none",
                    help="masking of the search_end_to_end leg on BOTH sides (the reference CLI is run with the same): none = --qmask none "
                         "--dbmask none (the round-1 figure), dust = the reference's default (DB masked on the device, queries on host threads)")
    ap.add_argument("--ref-search-queries", type=int, default=4096,
                    help="queries the reference CLI searches against the full DB (0 = skip; its index build takes ~1 min)")
    ap.add_argument("--dir-budget-gb", type=float, default=0.0)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default): every rank plays its own --queries queries (per-GPU work fixed); strong: the job of --queries "
                         "queries is cut into one contiguous block per rank (total work fixed)")
    ap.add_argument("--gather", choices=["async", "sync"], default="async",
                    help="N > 1: the final gather of hit records + CIGAR run words per step -- async: one fixed-capacity non-blocking "
                         "collective that overlaps the next step's kernels (sharding.FixedGather); sync: counts first, then padded payload")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="torch.distributed backend (nccl = RCCL; gloo: the one-GPU tests)")
    ap.add_argument("--single-rank-dist", action="store_true",
                    help="N = 1 only: initialise torch.distributed with ONE rank and run the N > 1 code path on it -- export, the (asynchronous) gather, "
                         "gather_check, --dry-collectives.  With --backend nccl that is RCCL executing every collective of the path on a box that has "
                         "one GPU (RCCL refuses two ranks on one device); the collectives are degenerate, the API usage is the real one")
    ap.add_argument("--dry-collectives", action="store_true",
                    help="N > 1: run ONLY the gather path (sharding.gather_results and FixedGather, incl. a forced overflow step) on tiny "
                         "tensors -- no database, no kernels -- and print one JSON line: tells a collective failure from a kernel failure")
    return ap.parse_args()


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    have_gpu = torch.cuda.is_available()
    if not have_gpu and not (a.dry_collectives and a.backend == "gloo"):
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")      # (the gloo dry run of the gather path is the one thing a CPU box can do)
    dev_index = local_rank % torch.cuda.device_count() if have_gpu else 0     # (several ranks may share a GPU: the one-GPU tests)
    if have_gpu:
        torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index) if have_gpu else torch.device("cpu")
    dist = None
    multi = world > 1 or (a.single_rank_dist and world == 1)      # the sharded code path (world == 1 only with --single-rank-dist)
    if multi:
        if world == 1:
            for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29651"), ("RANK", "0"), ("WORLD_SIZE", "1")):
                os.environ.setdefault(k, v)
        import torch.distributed as dist
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)     # RCCL
        else:
            dist.init_process_group("gloo")

    from vsearch_amd import sharding

    # what the collectives run on (VERDICT r03 "next" 8: the line of an N > 1 run describes its own transport)
    rccl = None
    if multi:
        try:
            nccl_version = ".".join(str(x) for x in torch.cuda.nccl.version()) if a.backend == "nccl" else None
        except Exception as e:                              # (never lose the run over a version string)
            nccl_version = repr(e)
        ident = torch.tensor([rank, dev_index, torch.cuda.device_count() if have_gpu else 0], dtype=torch.int64, device=(dev if a.backend == "nccl" else torch.device("cpu")))
        idents = [torch.zeros_like(ident) for _ in range(world)]
        dist.all_gather(idents, ident)                      # the first collective of the job: a transport problem shows HERE
        if a.backend == "nccl":
            torch.cuda.synchronize()
        _flush_c_stdio()                                    # (RCCL's version banner -- NCCL_DEBUG=VERSION on these boxes -- sits in the C stdout buffer:
                                                            #  out NOW, so that rank 0's JSON line is the last line of its stdout)
        rccl = {"world": world, "backend": a.backend, "nccl_version": nccl_version, "torch": torch.__version__,
                "hip": getattr(torch.version, "hip", None),
                "device_of_rank": [int(t[1]) for t in idents], "visible_devices_of_rank": [int(t[2]) for t in idents],
                "gpu": torch.cuda.get_device_name(dev_index) if have_gpu else None,
                "env": {k: os.environ.get(k) for k in ("HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_DEBUG", "MASTER_ADDR", "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES")}}
    if a.dry_collectives:
        if not multi:
            raise SystemExit("--dry-collectives needs N > 1 ranks (or --single-rank-dist)")
        res = dry_collectives(dist, sharding, dev if a.backend == "nccl" else torch.device("cpu"), rank, world)
        if rank == 0:
            _flush_c_stdio()
            print(json.dumps({"dry_collectives": res, "rccl": rccl, "n_gpus": world}), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    from vsearch_amd import Aligner, SequenceSet, workload

    # ---- synthetic inputs, generated in HBM.  Weak scaling: the job has world x a.queries queries, rank r owns the
    # contiguous block sharding.shard_queries() gives it (generated here from its own seed) and a replica of the DB
    t0 = time.time()
    db_ascii, db_off, db_len, fam = workload.make_family_db(a.db, a.dlen, seed=17, device=dev)
    strong = a.scaling == "strong" and world > 1
    if strong:
        # the whole job is generated identically on every rank (same seeds); a rank keeps the pairs of its block of queries
        q_ascii, q_off, q_len, src = workload.make_queries(db_ascii, db_off, db_len, a.queries, a.qlen, seed=11, device=dev)
        qidx, tidx = workload.family_candidates(src, fam, per_query=a.cands, seed=5)
        q_lo, q_hi = sharding.shard_queries(a.queries, world, rank)
        keep = (qidx >= q_lo) & (qidx < q_hi)
        qidx, tidx = qidx[keep], tidx[keep]
    else:
        q_lo, q_hi = sharding.shard_queries(world * a.queries, world, rank)
        q_ascii, q_off, q_len, src = workload.make_queries(db_ascii, db_off, db_len, q_hi - q_lo, a.qlen,
                                                           seed=11 + 1000 * rank, device=dev)
        qidx, tidx = workload.family_candidates(src, fam, per_query=a.cands, seed=5 + rank)
    torch.cuda.synchronize()
    t_gen = time.time() - t0

    al = Aligner(device=dev_index)
    T = SequenceSet(al, blob=db_ascii.numel(), offsets=db_off, lengths=db_len, device_ptr=db_ascii.data_ptr())
    Q = SequenceSet(al, blob=q_ascii.numel(), offsets=q_off, lengths=q_len, device_ptr=q_ascii.data_ptr())
    t0 = time.time()
    plan = al.plan(Q, T, qidx, tidx, dir_budget_bytes=int(a.dir_budget_gb * (1 << 30)))
    t_plan = time.time() - t0             # host: pairs -> wavefront tasks, task upload, checkpoint buffer (one-time hipMalloc)
    n_pairs = len(qidx)
    cells = int((q_len[qidx].astype(np.int64) * db_len[tidx].astype(np.int64)).sum())
    scratch = {}
    gathered = None
    fixed = sharding.FixedGather(dist, dst=0) if (multi and a.gather == "async") else None
    gloo = multi and a.backend == "gloo"

    def step():
        nonlocal gathered
        plan.run()
        tm = plan.sync()
        if multi:
            # the only collective on the path: final gather of the hit records and the CIGAR run words over RCCL/xGMI, to rank 0 (which
            # would write the output) -- the functions the world-2 tests drive (vsearch_amd/sharding.py).  async: the collective of this
            # step is posted and the PREVIOUS step's is collected, so it travels while the next step's kernels run
            rec, runs = sharding.export_records(plan, n_pairs, dev, scratch, host=gloo)
            if fixed is not None:
                # post step k, collect step k - 1 (a step that does not fit on SOME rank is redone synchronously by ALL ranks inside collect;
                # if the collectives never finish beside the next step's kernels ALL ranks agree -- through the same all-reduce -- to collect
                # at once from the same step on: FixedGather.step)
                for _, got in fixed.step(rec, runs):
                    gathered = got
            else:
                gathered = sharding.gather_results(rec, runs, dist, dst=0)
        return tm

    def drain():
        nonlocal gathered
        if fixed is not None:
            for _, got in fixed.drain():
                gathered = got

    def barrier():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    drain()
    barrier()
    t0 = time.perf_counter()
    fwd_ms = tb_ms = tot_ms = 0.0
    fwd_launches = 0
    tm = None
    for _ in range(a.steps):
        tm = step()
        fwd_ms += tm.forward_ms
        tb_ms += tm.traceback_ms
        tot_ms += tm.total_ms
        fwd_launches += tm.forward_launches
    drain()                                                # the last step's gather belongs to the timed region
    barrier()
    elapsed = time.perf_counter() - t0
    job_cells, job_pairs = cells, n_pairs
    per_rank_ms = None
    if multi:
        cdev = torch.device("cpu") if gloo else dev
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        tall = [torch.zeros_like(tmax) for _ in range(world)]
        dist.all_gather(tall, tmax)                          # every rank's own clock: a straggler shows in the line
        per_rank_ms = [round(float(t.item()) / a.steps * 1e3, 3) for t in tall]
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        tot = torch.tensor([cells, n_pairs], dtype=torch.int64, device=cdev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)           # what ALL ranks processed per step
        job_cells, job_pairs = int(tot[0]), int(tot[1])

    if rank != 0:
        if multi:
            dist.barrier()
            dist.destroy_process_group()
        return

    gather_check = None
    if multi and world == 1:
        # --single-rank-dist: the one rank's block must come back unchanged through the collective
        rec_all, runs_all, counts = gathered
        mine = sharding.decode_records(rec_all[:n_pairs])
        own = sharding.decode_records(scratch["rec"])
        gather_check = bool(list(counts) == [n_pairs] and np.array_equal(mine["score"], own["score"]) and np.array_equal(mine["run_off"], own["run_off"])
                            and np.array_equal(mine["nruns"], own["nruns"]) and int(runs_all.numel()) >= int(own["nruns"].sum()))
    elif world > 1:
        # rank 0 holds every rank's pairs: its own block must come back unchanged, offsets of the others rebased past it
        rec_all, runs_all, counts = gathered
        mine = sharding.decode_records(rec_all[:n_pairs])
        own = sharding.decode_records(scratch["rec"])
        other = sharding.decode_records(rec_all[counts[0]:counts[0] + counts[1]])
        n_runs0 = int(own["nruns"].sum())
        gather_check = bool(counts[0] == n_pairs and sum(counts) == job_pairs and np.array_equal(mine["score"], own["score"])
                            and np.array_equal(mine["run_off"], own["run_off"])
                            and int(other["run_off"][other["nruns"] > 0].min()) >= n_runs0
                            and int(runs_all.numel()) >= n_runs0 + int(other["nruns"].sum()))

    # ---- results on the host (PCIe + malloc'd arrays + CIGAR text), one plan, informational ----
    t0 = time.perf_counter()
    res = plan.fetch()
    t_fetch = time.perf_counter() - t0

    total_cells = job_cells * a.steps
    value = total_cells / elapsed / 1e9
    ms_per_step = elapsed / a.steps * 1e3
    fwd_avg_ms = fwd_ms / max(1, fwd_launches)
    cells_per_launch = cells * a.steps / max(1, fwd_launches)
    info = plan.describe()
    tilted = 2 * info["tasks_tilted"] >= info["tasks"]
    max3 = 2 * info.get("tasks_max3", 0) >= info["tasks"]
    PER_LANE_ROW = row_body_instructions(info)
    lane_rows_per_launch = cells_per_launch / 2.0
    achieved = lane_rows_per_launch * PER_LANE_ROW / (fwd_avg_ms * 1e-3) / 1e12      # T lane-instructions/s of the recurrence
    pmc, pmc_note = counters_from_profile(a, world)
    out = {
        "metric": "GCUPS (useful DP cells/s of the search16 global-alignment path: DP + traceback + CIGAR)",
        "value": round(value, 2),
        "unit": "GCUPS",
        "pairs_per_s": round(job_pairs * a.steps / elapsed, 1),
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": "int16", "data": "synthetic",
        "config": {
            "workload": f"usearch_global candidate batch: {a.queries} x {a.qlen} bp queries vs {a.db} x {a.dlen} bp "
                        f"family-structured DB (device-resident), {a.cands} candidates/query = {n_pairs} pairs/GPU/step, "
                        f"--id 0.9 shape, default scoring ({baseline_label(a)})",
            "candidates": "synthetic for `value`: source member + 7 same-family members (what the k-mer stage yields on this DB); "
                          "search_end_to_end runs the real device k-mer stage (vsx_kmer.hip) in front of the aligner",
            "parallelism": (f"query-sharded x{world} (sharding.shard_queries, {'strong: one job of ' + str(a.queries) + ' queries cut into blocks' if strong else 'weak: ' + str(a.queries) + ' queries per rank'}), "
                            f"DB replicated, one {a.gather} gather of records + CIGAR runs per step over {a.backend}")
                           if multi else "single GPU",
            "cells_per_step_per_gpu": cells,
        },
        "roofline": {
            "kernel": (f"vsx_forward_kernel<{info['rows_dominant']},true,false,true,true,{'true' if max3 else 'false'},{sparse_nq(info)},false,{'true' if info['rows_dominant'] * 16 >= a.qlen and sparse_nq(info) == 1 else 'false'}>" if tilted
                       else f"vsx_forward_kernel<{info['rows_dominant']},true,...>"),
            "bound": "valu-issue",
            "achieved": round(achieved, 3),
            "peak": round(PEAK_LANE_TOPS, 2),
            "peak_clock_ghz": 2.4,
            "effective_clock_ghz_profiled": (round(pmc["forward"]["sq"]["effective_clock_ghz"], 3) if pmc and pmc["forward"].get("sq", {}).get("effective_clock_ghz")
                                             else ("2.18 (r05: GRBM_GUI_ACTIVE 3.99e8 / 8 XCDs / 22.9 ms in the PMC pass): `peak` is the nominal 2.4 GHz and "
                                                   "overstates the ceiling of a sustained launch by about 9 %" if pmc else None)),
            "unit": "T lane-instructions/s (VALU issue: every instruction of the mix takes one ~4-cycle slot)",
            "frac": round(achieved / PEAK_LANE_TOPS, 4),
            "frac_of_measured_ceiling": round(achieved / (PEAK_LANE_TOPS * MEASURED_CEILING), 4),
            "row_body_instructions_per_lane_row": round(PER_LANE_ROW, 3),
            "arithmetic": "tilted+max3" if max3 else ("tilted" if tilted else "plain"),
            "plan": info,
            "kernel_ms_avg": round(fwd_avg_ms, 3),
            "kernel_launches": fwd_launches,
            "kernel_gcups": round(cells_per_launch / (fwd_avg_ms * 1e-3) / 1e9, 1),
            "traffic": pmc["forward"]["hbm_bytes_per_launch"] if pmc else None,
            "hbm_algorithmic_bytes_per_launch": int(tm.dir_bytes / max(1, tm.forward_launches)),
            "hbm_algorithmic_GBps": round(tm.dir_bytes / max(1, tm.forward_launches) / (fwd_avg_ms * 1e-3) / 1e9, 1),
            "not_a_bound": {"survey_8d_ops_per_cell": 15,
                            "note": "the reference's per-cell op count incl. 4 direction compares + min/max; this kernel does not execute those "
                                    "per cell (DESIGN.md 4.1), so it is not a lower bound here and carries no frac"},
        },
        "kernel_split_ms_per_step": {"forward": round(fwd_ms / a.steps, 3), "traceback": round(tb_ms / a.steps, 3),
                                     "cigar_text_and_rest": round((tot_ms - fwd_ms - tb_ms) / a.steps, 3)},
        "plan_s": round(t_plan, 3),
        "fetch_s": round(t_fetch, 3),
        "value_incl_fetch": round(cells / (ms_per_step * 1e-3 + t_fetch) / 1e9, 2),
        "gen_s": round(t_gen, 2),
    }
    if pmc:
        sq = pmc["forward"].get("sq", {})
        tb = pmc.get("traceback", {})
        block = {"kernel_source_sha": pmc["kernel_source_sha"],
                 "forward": {k: pmc["forward"].get(k) for k in ("hbm_bytes_per_launch", "hbm_read_bytes_per_launch", "hbm_write_bytes_per_launch")},
                 "traceback": {"hbm_bytes_per_launch": tb.get("hbm_bytes_per_launch"), "SQ_INSTS_VALU": tb.get("sq", {}).get("SQ_INSTS_VALU"),
                               "SQ_WAIT_ANY_over_WAVE_CYCLES": round(tb["sq"]["SQ_WAIT_ANY"] / tb["sq"]["SQ_WAVE_CYCLES"], 3)
                               if tb.get("sq", {}).get("SQ_WAVE_CYCLES") else None,
                               "kernel": tb.get("kernel"),
                               "lines_per_launch": int(tb["hbm_read_bytes_per_launch"] / 128) if tb.get("hbm_read_bytes_per_launch") else None,
                               "line_rate_floor_ms": round(tb["hbm_read_bytes_per_launch"] / 128 / 45e9 * 1e3, 3) if tb.get("hbm_read_bytes_per_launch") else None,
                               "note": "r05: the row checkpoints of a tile go HBM -> LDS without passing registers (R = 14, 16, >= 26; DESIGN 4.2).  r04: every partial-line checkpoint read costs HBM a whole 128-byte line (FETCH_SIZE calibrated on the kernel's own "
                                       "pattern, profiles/r04/r04a_fetch_calibration.txt; 45 G lines/s); the position-synchronous kernel keeps the lanes of a "
                                       "task on the same lines (5.05 pairs per fetched tile instead of 2.95) and issues 15 instead of 28 instructions per two "
                                       "recomputed cells: it is VALU-issue bound in the recompute (61 % of a wave's time), latency bound in loads and walk"},
                 "forward_sq": {k: sq.get(k) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES",
                                                       "SQ_WAIT_ANY", "GRBM_GUI_ACTIVE")}}
        if sq.get("SQ_INSTS_VALU"):
            # measured VALU issue occupancy of the DP kernel: executed wave-instructions (SQ_INSTS_VALU, PMC) x the CALIBRATED issue cost of
            # this mix (CPI_CALIBRATED, micro-benchmark on the loop's own row body) over the SIMD-cycles of the launch measured in THIS run
            block["valu_issue_occupancy"] = round(sq["SQ_INSTS_VALU"] * CPI_CALIBRATED / (256 * 4 * fwd_avg_ms * 1e-3 * 2.4e9), 3)
            block["cycles_per_instruction_calibrated"] = CPI_CALIBRATED
            block["valu_instructions_per_lane_row"] = round(sq["SQ_INSTS_VALU"] * 64 / (cells_per_launch / 2.0), 2)
        out["roofline"]["pmc"] = block
    else:
        out["roofline"]["traffic_note"] = pmc_note
    if gather_check is not None:
        out["gather_check"] = gather_check
    if rccl is not None:
        rccl["per_rank_ms_per_step"] = per_rank_ms
        rccl["gather"] = a.gather
        if fixed is not None:
            rccl["fixed_gather_sync_steps"] = fixed.sync_steps
            rccl["fixed_gather_overlap"] = fixed.stats()       # rank 0's view: did step k's collective finish while step k + 1 computed?
        out["rccl"] = rccl

    if not a.kernels_only and world == 1:
        # ---- end to end through the one-call entry (host index arrays in, host result arrays + CIGAR text out) ----
        try:
            out["end_to_end"] = end_to_end(a, al, Q, T, qidx, tidx, cells, res)
            out["value_end_to_end"] = out["end_to_end"]["value"]
        except Exception as e:
            out["end_to_end"] = {"error": repr(e)}

    # ---- CPU baseline: the reference's own SSE2 search16 (oracle/_ref, built from /root/reference) on a
    #      bounded sample of the SAME workload; falls back to the scalar port if _ref was not shipped ----
    if not a.no_cpu and not a.kernels_only and world == 1:          # rank 0 at N = 1 only: the other ranks would idle in the barrier
        try:
            out["cpu_baseline"] = cpu_baseline(a, db_ascii, db_off, db_len, q_ascii, q_off, q_len, qidx, tidx, res)
            if "value_end_to_end" in out and out["cpu_baseline"].get("value"):
                out["end_to_end"]["vs_cpu_baseline"] = round(out["value_end_to_end"] / out["cpu_baseline"]["value"], 1)
        except Exception as e:  # the baseline is a side measurement: never lose the bench line over it
            out["cpu_baseline"] = {"error": repr(e)}

    if not a.kernels_only and not a.no_search and world == 1:
        plan.close()
        try:
            out["search_end_to_end"] = search_end_to_end(a, al, db_ascii, db_off, db_len, q_ascii, q_off, q_len)
        except Exception as e:
            import traceback
            out["search_end_to_end"] = {"error": repr(e), "traceback": traceback.format_exc()[-1500:]}
    if not a.kernels_only and not a.no_shapes and world == 1:
        try:
            plan.close()
        except Exception:
            pass
        try:
            out["shapes"] = shapes_leg(a, al, dev)
        except Exception as e:
            out["shapes"] = {"error": repr(e)}
    _flush_c_stdio()
    print(json.dumps(out), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


# the pair shapes of BASELINE configs 5 / 3 / 4 (VERDICT r03 "next" 4: driver-timed instead of builder-only): 100 k queries x 8 family
# candidates each, kernels only (DP + traceback + CIGAR text, results resident), same generator and seeds as the main workload
# config 5 is 10 M x 150 bp queries vs 5 M x 1 kbp: its pair shape is 150 x 1000.  `short_150x300` is NOT a BASELINE shape
# (r02-r04 printed it as "config5": VERDICT r04 missing 1); it stays as the short x short stress of the per-step overheads.
# `config4_400x400_dense32`: config 4 is dense all-vs-all -- every query meets thousands of targets -- so next to the 8-candidate form the same
# 800 k pairs are also run as 25 000 queries x 32 candidates: there the pair-profile classes engage (four tasks of one query share a
# pair-indexed profile), as they do inside --allpairs_global itself (bench_allpairs.py).
SHAPES = (("config5_150x1000", 150, 1000, 1_000_000, 100_000, 8), ("config3_300x300", 300, 300, 400_000, 100_000, 8),
          ("config4_400x400", 400, 400, 300_000, 100_000, 8), ("config4_400x400_dense32", 400, 400, 300_000, 25_000, 32),
          ("short_150x300", 150, 300, 400_000, 100_000, 8))


def shapes_leg(a, al, dev, steps=3):
    import types
    from vsearch_amd import SequenceSet, workload
    out = {}
    for name, qlen, dlen, dbn, nq, ncand in SHAPES:
        db_ascii, db_off, db_len, fam = workload.make_family_db(dbn, dlen, seed=17, device=dev)
        q_ascii, q_off, q_len, src = workload.make_queries(db_ascii, db_off, db_len, nq, qlen, seed=11, device=dev)
        qidx, tidx = workload.family_candidates(src, fam, per_query=ncand, seed=5)
        torch.cuda.synchronize()
        T = SequenceSet(al, blob=db_ascii.numel(), offsets=db_off, lengths=db_len, device_ptr=db_ascii.data_ptr())
        Q = SequenceSet(al, blob=q_ascii.numel(), offsets=q_off, lengths=q_len, device_ptr=q_ascii.data_ptr())
        plan = al.plan(Q, T, qidx, tidx)
        cells = int((q_len[qidx].astype(np.int64) * db_len[tidx].astype(np.int64)).sum())
        plan.run()
        plan.sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fwd = tb = tot = 0.0
        for _ in range(steps):
            plan.run()
            tm = plan.sync()
            fwd += tm.forward_ms; tb += tm.traceback_ms; tot += tm.total_ms
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        info = plan.describe()
        entry = {"value": round(cells * steps / elapsed / 1e9, 2), "unit": "GCUPS", "ms_per_step": round(elapsed / steps * 1e3, 3), "steps": steps,
                 "pairs_per_step": len(qidx), "queries": nq, "candidates_per_query": ncand, "rows_per_lane": info["rows_dominant"],
                 "tasks_pair_profile": info.get("tasks_pair", 0),
                 "kernel_split_ms_per_step": {"forward": round(fwd / steps, 3), "traceback": round(tb / steps, 3),
                                              "cigar_text_and_rest": round((tot - fwd - tb) / steps, 3)}}
        if not a.no_cpu:
            try:
                res = plan.fetch()
                ns = types.SimpleNamespace(cpu_pairs=40_000, cands=ncand, cpu_threads=a.cpu_threads)
                cb = cpu_baseline(ns, db_ascii, db_off, db_len, q_ascii, q_off, q_len, qidx, tidx, res)
                entry["parity_all_fields_match"] = cb.get("parity_all_fields_match", cb.get("parity_match"))
                entry["parity_sample_pairs"] = 40_000 if "parity_all_fields_match" in cb else 2000
                entry["cpu_reference_GCUPS"] = cb.get("value")
            except Exception as e:
                entry["parity_error"] = repr(e)
        out[name] = entry
        plan.close()
        Q.close(); T.close()
        del db_ascii, q_ascii
        torch.cuda.empty_cache()
    return out


def dry_collectives(dist, sharding, device, rank, world):
    """The gather path alone, on tiny synthetic records: the synchronous gather, the asynchronous fixed-capacity form with two steps in
    flight, and a step that overflows the fixed capacity on ONE rank (every rank then redoes it synchronously).  Each rank builds every
    rank's payload deterministically, so every receiver checks what it got."""
    def payload(r, step, scale=1):
        n = (5 + 3 * r + step) * scale
        nr = np.arange(n, dtype=np.uint32) % 3 + 1
        off = np.concatenate([[0], np.cumsum(nr[:-1])]).astype(np.uint64)
        rec = sharding.pack_records(np.arange(n) + 100 * r, nr * 2, nr, nr * 0, nr * 0, nr, off)
        runs = (np.arange(int(nr.sum()), dtype=np.int32) * 4 + r)
        return torch.from_numpy(rec).to(device), torch.from_numpy(runs).to(device)

    def expect(step, scale_of_rank):
        recs, runs, counts, base = [], [], [], 0
        for r in range(world):
            rc, rn = payload(r, step, scale_of_rank(r))
            rc = rc.clone()
            if rc.shape[0]:
                rc.view(torch.int64).view(-1, 3)[:, 2] += base
            recs.append(rc); runs.append(rn); counts.append(int(rc.shape[0])); base += int(rn.numel())
        return torch.cat(recs), torch.cat(runs), counts

    checks = {}
    rec, runs = payload(rank, 0)
    got = sharding.gather_results(rec, runs, dist, dst=0)
    if rank == 0:
        e = expect(0, lambda r: 1)
        checks["gather_results_to_rank0"] = bool(torch.equal(got[0], e[0]) and torch.equal(got[1], e[1]) and got[2] == e[2])
    fg = sharding.FixedGather(dist, dst=0)
    scales = [lambda r: 1, lambda r: 1, lambda r: (40 if r == world - 1 else 1), lambda r: 1]       # step 2 overflows on the last rank only
    tickets = []
    ok = True
    for step, sc in enumerate(scales):
        rec, runs = payload(rank, step, sc(rank))
        tickets.append(fg.post(rec, runs))
        if step >= 1:
            got = fg.collect(tickets[step - 1])
            if rank == 0:
                e = expect(step - 1, scales[step - 1])
                ok = ok and bool(torch.equal(got[0], e[0]) and torch.equal(got[1], e[1]) and got[2] == e[2])
    got = fg.collect(tickets[-1])
    if rank == 0:
        e = expect(len(scales) - 1, scales[-1])
        ok = ok and bool(torch.equal(got[0], e[0]) and torch.equal(got[1], e[1]) and got[2] == e[2])
        checks["fixed_gather_async_incl_one_rank_overflow"] = ok
    checks["fixed_gather_sync_steps"] = fg.sync_steps
    st = fg.stats()                 # (r05) every collected step is an overlap sample: finished before the collect, or waited for
    checks["fixed_gather_overlap_samples"] = st["collects"] == st["finished_before_collect"] + st["waited_for"] == len(scales)
    # r06: the ranks' OWN overlap samples disagree -- rank 0 always had to wait, the others never -- and steps
    # overflow on one rank only: before the degrade decision became collective, rank 0 flipped alone and ran step 3's synchronous redo
    # at another position of the collective sequence than its peers.  Now every rank flips while collecting the same ticket; the loop
    # is FixedGather.step / drain, the one the timed steps use.  Step 3 overflows on the last rank (still pipelined), step 6 on rank 0
    # (already degraded).
    fg = sharding.FixedGather(dist, dst=0)
    fg.force_sample = (rank != 0)
    scales = [lambda r: 1, lambda r: 1, lambda r: 1, lambda r: (40 if r == world - 1 else 1), lambda r: 1, lambda r: 1,
              lambda r: (400 if r == 0 else 1), lambda r: 1]
    done, ok, flipped_at = [], True, None
    for step, sc in enumerate(scales):
        rec, runs = payload(rank, step, sc(rank))
        was = fg.degraded
        done += fg.step(rec, runs)
        if fg.degraded and not was:
            flipped_at = step
    done += fg.drain()
    ok = [k for k, _ in done] == list(range(len(scales)))
    if rank == 0:
        for k, got in done:
            e = expect(k, scales[k])
            ok = ok and bool(torch.equal(got[0], e[0]) and torch.equal(got[1], e[1]) and got[2] == e[2])
    # every rank must have flipped in the same step and redone the same two steps
    mine = torch.tensor([int(ok), -1 if flipped_at is None else flipped_at, fg.sync_steps], dtype=torch.int64, device=device)
    views = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(views, mine)
    checks["fixed_gather_mixed_votes_one_rank_overflows"] = all(int(v[0]) == 1 and int(v[1]) == int(views[0][1]) >= 0 and int(v[2]) == 2 for v in views)
    checks["fixed_gather_degraded_at_step"] = int(views[0][1])
    return checks


def end_to_end(a, al, Q, T, qidx, tidx, cells, res):
    """vsx_align_pairs (plan + kernels + fetch, pipelined slices inside the library) on the step's pair list, steady state:
    one untimed call (pools, pinned staging), then a.e2e_calls timed calls back to back."""
    r = al.align_pairs_raw(Q, T, qidx, tidx)
    n = len(r)
    same = bool(np.array_equal(r.score, res.score) and np.array_equal(r.aligned, res.aligned) and np.array_equal(r.matches, res.matches)
                and np.array_equal(r.gaps, res.gaps) and all(r.cigar(k) == res.cigar[k] for k in range(0, n, max(1, n // 5000))))
    text_bytes = r.cigar_bytes
    r.close()
    calls = max(1, a.e2e_calls)
    times = []
    thr0 = cpu_throttle()
    for _ in range(calls):
        t0 = time.perf_counter()
        r = al.align_pairs_raw(Q, T, qidx, tidx)
        times.append(time.perf_counter() - t0)
        r.close()
    thr = throttle_delta(thr0)
    avg = sum(times) / calls
    med = float(np.median(times))
    return {"value": round(cells / avg / 1e9, 2), "unit": "GCUPS", "pairs_per_s": round(n / avg, 1),
            "ms_per_call": round(avg * 1e3, 2), "ms_per_call_min": round(min(times) * 1e3, 2), "ms_per_call_median": round(med * 1e3, 2),
            "value_median": round(cells / med / 1e9, 2), "ms_calls": [round(t * 1e3, 2) for t in times], "calls": calls,
            "cpu_throttle_during_calls": thr,
            "what": "vsx_align_pairs: host pair list -> planning, DP + traceback + CIGAR-text kernels, PCIe, malloc'd result arrays and "
                    "CIGAR strings on the host; slices of the list pipelined inside the library; pools warm",
            "cigar_text_bytes": int(text_bytes), "equals_plan_results": same}


def _write_fasta(path, blob, off, ln, prefix):
    with open(path, "wb") as f:
        parts = []
        for i in range(len(ln)):
            o = int(off[i])
            parts.append(b">%s%d\n%s\n" % (prefix, i, blob[o:o + int(ln[i])]))
            if len(parts) >= 65536:
                f.write(b"".join(parts))
                parts = []
        f.write(b"".join(parts))


def search_end_to_end(a, al, db_ascii, db_off, db_len, q_ascii, q_off, q_len):
    """vsx_search_batch = --usearch_global --id 0.9 (defaults: maxaccepts 1, maxrejects 32, wordlength 8) for ALL queries of the
    step against the DB: device k-mer counting + candidate ranking + the reference's accept/reject loop + alignments + hits.
    The reference CLI (oracle/_ref/vsearch_ref, same options, --qmask/--dbmask none, all usable cores) searches the first
    --ref-search-queries of the same queries against the same DB file; its search phase is timed from its own progress
    output (the 'Searching' prompt appears when the phase starts), index construction excluded."""
    from vsearch_amd import _lib
    from vsearch_amd._lib import check
    lib = _lib.load()
    db_blob = db_ascii.cpu().numpy().tobytes()
    q_blob = q_ascii.cpu().numpy().tobytes()

    def vp(arr):
        return arr.ctypes.data_as(C.c_void_p)

    o = _lib.SearchOpts()
    lib.vsx_search_opts_default(C.byref(o))
    o.id = 0.9
    o.soft_mask = 2 if a.search_mask == "dust" else 0
    h = C.c_void_p()
    t0 = time.perf_counter()
    check(lib.vsx_searcher_create(al.h, C.byref(h), C.byref(o), len(db_len), C.cast(C.c_char_p(db_blob), C.c_void_p),
                                  len(db_blob), vp(db_off), vp(db_len)), "vsx_searcher_create")
    t_create = time.perf_counter() - t0
    nq = len(q_len)
    out = {}
    try:
        secs = []
        hits = None
        thr0 = None
        for rep in range(max(2, int(os.environ.get("VSX_BENCH_SEARCH_REPS", "6")))):   # the first call pays the one-time index build and scratch-pool hipMalloc
            if hits is not None:
                lib.vsx_hits_free(C.byref(hits))
            hits = _lib.Hits()
            if rep == 1:
                thr0 = cpu_throttle()                       # (the later, warm calls)
            t0 = time.perf_counter()
            check(lib.vsx_search_batch(h, nq, C.cast(C.c_char_p(q_blob), C.c_void_p), len(q_blob), vp(q_off), vp(q_len),
                                       C.byref(hits)), "vsx_search_batch")
            secs.append(time.perf_counter() - t0)
        thr = throttle_delta(thr0)
        best = min(secs[1:])
        med = float(np.median(secs[1:]))
        first = np.ctypeslib.as_array(hits.first, shape=(nq + 1,)).copy()
        # r06: the figure is the MEDIAN of the warm calls; the best call stands beside it
        out = {"queries": nq, "seconds": round(med, 4), "seconds_best": round(best, 4), "seconds_first_call": round(secs[0], 3), "seconds_later_calls": [round(x, 4) for x in secs[1:]], "seconds_median": round(med, 4),
               "queries_per_s": round(nq / med, 1), "queries_per_s_best": round(nq / best, 1),
               "cpu_throttle_during_later_calls": thr,
               "pairs_aligned": int(hits.pairs_aligned), "cells_aligned": int(hits.cells_aligned),
               "value": round(int(hits.cells_aligned) / med / 1e9, 2), "unit": "GCUPS (cells THIS dispatch aligns / wall; r05: lazy first batches -- a query's first batch is the accepts it still needs, so fewer cells than the reference's dispatch aligns for the same hits; VSX_SEARCH_LAZY=0 restores its batches of eight)",
               "hits": int(hits.n_hits), "queries_with_hit": int((first[1:] > first[:-1]).sum()),
               "seconds_kmer": round(hits.seconds_kmer, 3), "seconds_align": round(hits.seconds_align, 3),
               "searcher_create_s": round(t_create, 2), "masking": a.search_mask}
        nref = min(a.ref_search_queries, nq)
        ref_bin = os.path.join(ROOT, "oracle", "_ref", "vsearch_ref")
        if nref > 0 and not a.no_cpu and os.path.exists(ref_bin):
            harr = np.ctypeslib.as_array(hits.hit, shape=(max(1, int(hits.n_hits)),))[:int(hits.n_hits)]
            keep = np.nonzero((harr["query"] < nref) & (harr["accepted"] != 0))[0]
            cblob = C.string_at(hits.cigar_blob, int(hits.cigar_bytes)) if hits.cigar_bytes else b""
            # what --userfields query+target+id+caln prints for a hit (id with one decimal, commands/userfields: "%.1f")
            ours = set()
            for k in keep.tolist():
                hr = harr[k]                    # (not `h`: that is the searcher's handle, destroyed in the finally below)
                o0 = int(hr["cigar_off"])
                ours.add((int(hr["query"]), int(hr["target"]), "%.1f" % float(hr["id"]), cblob[o0:cblob.index(b"\0", o0)].decode()))
            try:                                            # (a 5 GB FASTA + the reference's index may not fit a box: never lose the search figures over it)
                out["reference_cli"] = reference_search(ref_bin, db_blob, db_off, db_len, q_blob, q_off, q_len, nref, ours, a.search_mask)
            except Exception as e:
                out["reference_cli"] = {"error": repr(e)}
            rq = out["reference_cli"].get("queries_per_s")
            if rq:
                out["vs_reference_cli"] = round(out["queries_per_s"] / rq, 1)
    finally:
        if hits is not None:
            lib.vsx_hits_free(C.byref(hits))
        lib.vsx_searcher_destroy(h)
    return out


def reference_search(ref_bin, db_blob, db_off, db_len, q_blob, q_off, q_len, nref, ours, masking="none"):
    threads = usable_cpus()
    with tempfile.TemporaryDirectory(prefix="vsxbench_") as tmp:
        dbf, qf, uo = (os.path.join(tmp, x) for x in ("db.fa", "q.fa", "u.txt"))
        _write_fasta(dbf, db_blob, db_off, db_len, b"t")
        _write_fasta(qf, q_blob, q_off[:nref], q_len[:nref], b"q")
        mask_args = ["--qmask", "none", "--dbmask", "none"] if masking == "none" else []       # no option = dust on both sides
        cmd = [ref_bin, "--usearch_global", qf, "--db", dbf, "--id", "0.9", "--threads", str(threads)] + mask_args + [
               "--userout", uo, "--userfields", "query+target+id+caln"]
        t_start = time.perf_counter()
        p = subprocess.Popen(cmd, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL)
        stamp = {}

        def watch():
            buf = b""
            while True:
                c = p.stderr.read(1)
                if not c:
                    break
                buf += c
                if "search_begin" not in stamp:
                    if buf.endswith(b"Searching"):
                        stamp["search_begin"] = time.perf_counter()
                        buf = b""
                elif "search_end" not in stamp and buf.endswith(b"100%"):
                    stamp["search_end"] = time.perf_counter()
        th = threading.Thread(target=watch)
        th.start()
        rc = p.wait()
        t_end = time.perf_counter()
        th.join()
        if rc != 0 or "search_begin" not in stamp:
            return {"error": f"vsearch_ref rc {rc}"}
        secs = stamp.get("search_end", t_end) - stamp["search_begin"]
        theirs = set()
        with open(uo) as f:
            for line in f:
                qn, tn, idv, caln = line.rstrip("\n").split("\t")
                theirs.add((int(qn[1:]), int(tn[1:]), idv, caln))
        return {"queries": nref, "threads": threads, "search_seconds": round(secs, 3), "queries_per_s": round(nref / secs, 1),
                "load_and_index_seconds": round(stamp["search_begin"] - t_start, 1),
                "what": "vsearch_ref --usearch_global --id 0.9 " + " ".join(mask_args) + (" " if mask_args else "(default dust masking) ") +
                        "search phase only (from its 'Searching' prompt to the '100%' that ends it; DB masking and indexing are in "
                        "load_and_index_seconds), full DB",
                "hits": len(theirs), "same_hits_as_vsx": bool(theirs == ours),
                "same_pairs_as_vsx": bool({x[:2] for x in theirs} == {x[:2] for x in ours}),
                "compared_fields": "query+target+id+caln (every --userout line of the sample as a tuple; set equality)"}


def _flush_c_stdio():
    """flush every C stdio stream of the process (fflush(NULL)): text written by native libraries with printf"""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:                              # noqa: BLE001
        pass


def baseline_label(a):
    """which BASELINE.json config the pair shape of THIS run is (derived from the arguments, not assumed)"""
    shape = (a.queries, a.qlen, a.db, a.dlen, a.cands)
    if shape == (100_000, 250, 1_000_000, 1000, 8):
        return "BASELINE config[1] at full size"
    if (a.qlen, a.db, a.dlen, a.cands) == (150, 5_000_000, 1000, 8):
        return f"BASELINE config[4]'s pair shape and DB; {a.queries} queries = {'one GPU of eight' if a.queries == 1_250_000 else 'a part'} of its 10 M"
    if (a.qlen, a.dlen, a.cands) == (250, 1000, 8) and a.queries == 1000 and a.db == 10_000:
        return "BASELINE config[0] (the plumbing case)"
    return f"not a BASELINE configuration: {a.queries} x {a.qlen} bp vs {a.db} x {a.dlen} bp, {a.cands} candidates"


def cpu_throttle():
    """cgroup v2 CPU throttling counters of this container: a quota'd box (cpu.max) stalls EVERY thread of the process for the rest of a
    100 ms period once the quota is spent -- a 5-10 ms hole in a 120 ms call that no kernel explains"""
    try:
        d = dict(ln.split() for ln in open("/sys/fs/cgroup/cpu.stat").read().splitlines())
        return {"nr_periods": int(d.get("nr_periods", 0)), "nr_throttled": int(d.get("nr_throttled", 0)), "throttled_usec": int(d.get("throttled_usec", 0)),
                "usage_usec": int(d.get("usage_usec", 0))}
    except Exception:
        return None


def throttle_delta(before):
    after = cpu_throttle()
    if before is None or after is None:
        return None
    return {k: after[k] - before[k] for k in before}


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
        ref_digest = ref.last_digest
        gpu_chk = int(res.score[:n].astype(np.int64).sum() + res.aligned[:n].astype(np.int64).sum()
                      + res.matches[:n].astype(np.int64).sum())
        # every field of every pair of the sample, CIGAR text included: an order-independent hash of (pair number, score, aligned,
        # matches, mismatches, gaps, CIGAR) computed by the reference driver over ITS results and over the GPU path's
        cig = [c.encode() for c in res.cigar[:n]]
        blob = b"\0".join(cig) + b"\0"
        off = np.zeros(n, np.uint64)
        if n > 1:
            off[1:] = np.cumsum(np.fromiter((len(c) + 1 for c in cig[:-1]), np.uint64, n - 1))
        gpu_digest = ref.digest_results(0, res.score[:n], res.aligned[:n], res.matches[:n], res.mismatches[:n], res.gaps[:n],
                                        blob, off)
        return {"value": round(c / secs / 1e9, 2), "unit": "GCUPS", "cores": threads, "kind": "reference",
                "sample": f"reference SSE2 search16 (oracle/_ref, -O3 -march=x86-64) on the first {n} pairs "
                          f"({len(uq)} queries x {a.cands}) of the same step, {threads} std::threads "
                          f"(= usable host CPUs: affinity {len(os.sched_getaffinity(0))}, cgroup quota applied; "
                          f"{os.cpu_count()} online), {secs:.2f} s wall",
                "single_thread_GCUPS": round(one_cells / one_s / 1e9, 2),
                "parity_checksum_match": bool(chk == gpu_chk),
                "parity_all_fields_match": bool(ref_digest == gpu_digest),
                "parity_all_fields_what": f"hash over (pair, score, aligned, matches, mismatches, gaps, CIGAR text) of the {n} sample pairs: "
                                          "reference SSE2 search16 vs the GPU path"}
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
