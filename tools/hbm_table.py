#!/usr/bin/env python3
"""Per-kernel HBM table of the bandwidth-bound kernels, each measured alone (EBCC_HIP_SLICES=1, 256 frames):
    python3 tools/hbm_table.py <kernel_trace.csv> <FETCH_SIZE counter csv> <WRITE_SIZE counter csv> <frames> <sha> > profiles/r03_hbm_kernels.json
The three CSVs come from three runs of the same deterministic command (tools/gpu/hbm_table.sh), so dispatch k of one
run is dispatch k of the others; a row of the table is one (kernel, launch geometry) pair and reports the dispatches
in which every frame of the batch was active (counter bytes within 10 % of the largest for that pair).
  algorithmic bytes: what the kernel has to move by its role (see ALGO below; per frame x frames)
  counter bytes:     FETCH_SIZE x 2 (gfx950 correction of the micro-architecture guide) + WRITE_SIZE, units of 1 KB
  achieved:          algorithmic bytes / duration; fractions of the 8.0 TB/s spec and of the 6.29 TB/s measured copy rate"""
import csv
import json
import re
import sys
from collections import defaultdict

H, W = 721, 1440
PIX = H * W
NY, NX = 736, 1440                               # padded SPIHT grid
RES = [(45, 23), (90, 46), (180, 91), (360, 181), (720, 361), (1440, 721)]   # (width, height) of resolution 0..5


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name).replace("ebcc::", "")
    m = re.match(r"([A-Za-z0-9_]+(<[^>(]*>)?)", name)
    return m.group(1) if m else name[:48]


def algo_bytes(kernel, rank, nkeys):
    """Bytes per frame the kernel must move by its role; `rank` = position of this launch geometry among the kernel's
    geometries, largest first (level 5 / level 3 of the residual transform first).  Kernel names are matched whole
    (k_rate is not k_rate_publish, k_residual_minmax is not k_residual_minmax_finish)."""
    k = re.sub(r"<.*", "", kernel)                       # name without template arguments
    targs = kernel[len(k):]
    if k == "k_scale_shift": return 2 * 4 * PIX, "read fp32 frame, write shifted fp32 samples"
    if k == "k_in_minmax": return 4 * PIX, "read fp32 frame"
    if k == "k_quantize": return 4 * PIX + 4 * PIX + PIX * 55 // 8, "read coefficients, write Q6 + 26 plane + sign + 28 suffix masks (1 bit each per sample)"
    if k == "k_j2k_level5_fin" and targs.startswith("<true"): return 8 * PIX, "whole top level: read the four bands (LL fp32, rest int32) and the fp32 frame for the statistics (field kept: + 4 B/sample write)"
    if k == "k_j2k_level5_fin":
        lv = 4 - rank if rank < 4 else 1                 # geometries of levels 4, 3, 2, 1 - largest first
        w, h = RES[lv]
        return 8 * w * h, f"whole level {lv}: read the four bands, write {w}x{h} fp32"
    if k == "k_finest_inv_use": return 4 * (NX * NY // 4) + 4 * 3 * (NX * NY // 4) + 8 * PIX, "finest residual level whole: LL read, the ordinals of three detail bands (+ coefficient and slot inside the prefix), frame and decoded field for the statistics"
    if k == "k_j2k_cols_fin": return 8 * PIX, "level 5 columns: read the level and the fp32 frame (field kept: + 4 B/sample write)"
    if k in ("k_j2k_rows", "k_j2k_cols"):
        lv = 5 - rank if rank < 5 else 1
        w, h = RES[lv]
        fin = "true>" in targs.replace(" ", "") and k == "k_j2k_cols" and targs.startswith("<false")
        if fin: return 4 * w * h + 4 * PIX, f"level {lv}: read the level, read the fp32 frame for the statistics (field kept: + 4 B/sample write)"
        return 8 * w * h, f"level {lv}: read + write {w}x{h} fp32"
    if k == "k_j2k_cols_fwd_top": return 8 * PIX, "level 5: read the fp32 frame, write the column-transformed level"
    if k == "k_pad_load": return 8 * PIX + 4 * NY * NX, "read frame + decoded field, write padded grid"
    if k == "k_residual_minmax": return 8 * PIX, "read frame + decoded field"
    if k in ("k_rows_fwd", "k_rows_inv", "k_cols_fwd", "k_cols_inv"):
        lv = rank                                  # 0 = full grid
        return 8 * (NX >> lv) * (NY >> lv), f"level {3 - lv}: read + write {(NX >> lv)}x{(NY >> lv)} fp32"
    if k == "k_cols_inv_stream": return 12 * 3 * (NX * NY // 4) + 4 * (NX * NY // 4) + 4 * NX * NY, "finest level columns: three detail bands from the bookkeeping (coefficient + two ordinals), LL read, grid written"
    if k == "k_rows_inv_use": return 4 * NX * NY + 8 * PIX, "read the grid, the frame and the decoded field (statistics only)"
    if k == "k_truncate": return 8 * NX * NY, "read fp32 grid, write int32 coefficients"
    if k == "k_descmax":
        # one launch per level, coarsest parents last: level l reads the 2^-l x 2^-l grid of coefficients (or of maxima) and writes a quarter of it twice
        n = (NX >> rank) * (NY >> rank)
        return 4 * n + 2 * 4 * (n // 4), f"level {rank}: read {NX >> rank}x{NY >> rank} int32, write the two maxima of its parents"
    if k == "k_reconstruct": return 16 * NX * NY, "read coefficient + two ordinals, write fp32 grid"
    if k == "k_distortion": return 4 * PIX, "read Q6"
    if k == "k_int_to_float": return 8 * NX * NY, "read int32, write fp32"
    # latency-bound or data-dependent kernels (k_rate, k_t1_resume, the j2k k_probe_init, k_rate_publish, the *_finish sweeps ...)
    # have no byte count by role: they are not rows of this table
    return None, ""


def load_counter(path, counter):
    out = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            out[int(r["Dispatch_Id"])] = (short(r["Kernel_Name"]), float(r["Counter_Value"]), int(r.get("Grid_Size", 0) or 0))
    return out


trace, fpath, wpath, frames, sha = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
dur = {}
for r in csv.DictReader(open(trace)):
    grid = int(r.get("Grid_Size_X", 0) or 0) * max(1, int(r.get("Grid_Size_Y", 1) or 1)) * max(1, int(r.get("Grid_Size_Z", 1) or 1))
    dur[int(r["Dispatch_Id"])] = (short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, grid)
fetch, write = load_counter(fpath, "FETCH_SIZE"), load_counter(wpath, "WRITE_SIZE")
groups = defaultdict(list)
mismatch = 0
for d, (name, us, grid) in dur.items():
    if d not in fetch or d not in write or fetch[d][0] != name or write[d][0] != name:
        mismatch += 1
        continue
    groups[(name, grid)].append((us, 2 * fetch[d][1] * 1024 + write[d][1] * 1024, 2 * fetch[d][1] * 1024, write[d][1] * 1024))
by_kernel = defaultdict(list)
for (name, grid) in groups:
    by_kernel[name].append(grid)
rows = []
for (name, grid), lst in groups.items():
    geoms = sorted(set(by_kernel[name]), reverse=True)
    a, what = algo_bytes(name, geoms.index(grid), len(geoms))
    if a is None:
        continue
    top = max(t[1] for t in lst)
    full = [t for t in lst if t[1] >= 0.9 * top] or lst
    us = sum(t[0] for t in full) / len(full)
    cb = sum(t[1] for t in full) / len(full)
    fb = sum(t[2] for t in full) / len(full)
    wb = sum(t[3] for t in full) / len(full)
    ab = a * frames
    # what the dispatches with all frames active have to move beyond the per-round minimum: the probe that keeps its field
    # writes it (the first probe of a search does), and the first truncation probe has the whole stream inside its prefix
    # (coefficient + slot of every significant detail sample: counted from the write / fetch split, not assumed)
    if name.startswith("k_j2k_level5_fin<true") and wb > 2 * PIX * frames:
        ab += 4 * PIX * frames
        what += " - these dispatches keep the field: 12 B/sample"
    gbps = ab / us / 1e3
    if grid < 65536:
        continue                                     # (a launch of a few hundred threads is not a bandwidth measurement)
    row = {"kernel": name, "grid_threads": grid, "what": what, "dispatches": len(lst), "dispatches_all_frames_active": len(full),
           "duration_us": round(us, 1), "algorithmic_bytes": ab, "counter_bytes": int(cb), "counter_fetch_bytes": int(fb), "counter_write_bytes": int(wb),
           "counter_over_algorithmic": round(cb / ab, 2),
           "achieved_GBps": round(gbps, 1), "frac_of_8000": round(gbps / 8000, 4), "frac_of_6290": round(gbps / 6290, 4)}
    if cb < 0.5 * ab or gbps > 6290:
        # the input was written by the kernel before and is (partly) still in the 256 MB Infinity Cache: not an HBM rate
        # (counter bytes below half the algorithmic ones, or a rate above what HBM can deliver)
        row["cache_served"] = True
        row["frac_of_8000"] = row["frac_of_6290"] = None
    rows.append(row)
rows.sort(key=lambda r: (-r["duration_us"] * r["dispatches"]))
print(json.dumps({"frames_per_dispatch": frames, "kernel_sources_sha": sha, "unmatched_dispatches": mismatch,
                  "note": "EBCC_HIP_SLICES=1: every kernel runs alone; durations from rocprofv3 --kernel-trace, bytes from separate --pmc FETCH_SIZE / WRITE_SIZE runs",
                  "kernels": rows}, indent=1))
