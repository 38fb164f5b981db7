#!/usr/bin/env python3
"""Condense bench/profile.sh's rocprofv3 CSVs into profiles/<tag>_rocprof_summary.md and profiles/pmc_traffic.json.

HBM bytes follow MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB-units of the TCC's fabric-side
request counters; on gfx950 FETCH_SIZE reports exactly half of a wide coalesced streaming read, so it is doubled.
"""
import csv
import glob
import json
import os
import sys

KERNEL = "nbody_force_f32"            # K1 (every ordered pair); switched to K1s' names below when the trace holds them
REDUCER = "nbody_reduce_update_f32"   # counted in the step's traffic: kernel_ms spans force + reducer, so must the bytes


def rows(pattern):
    for path in glob.glob(pattern, recursive=True):
        with open(path, newline="") as f:
            yield from csv.DictReader(f)


def main():
    global KERNEL, REDUCER
    out, tag = sys.argv[1], sys.argv[2]
    if any("nbody_force_sym_f32" in r.get("Name", "") for r in rows(f"{out}/stats/**/*kernel_stats.csv")):
        KERNEL, REDUCER = "nbody_force_sym_f32", "nbody_reduce_sym_f32"  # K1s: every unordered pair once
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lines = [f"# rocprofv3 summary `{tag}` (bench/profile.sh; raw CSVs under {out}, scratch)", ""]
    # 1. kernel stats
    stats = list(rows(f"{out}/stats/**/*kernel_stats.csv"))
    lines += ["## --kernel-trace --stats", "", "| kernel | calls | total ms | avg ms | % |", "|---|---|---|---|---|"]
    avg_ms = kernel = None
    reduce_total = force_total = 0.0
    for r in stats:
        name = r.get("Name", "")
        avg = float(r.get("AverageNs", 0)) / 1e6
        lines.append(f"| `{name[:90]}` | {r.get('Calls')} | {float(r.get('TotalDurationNs', 0)) / 1e6:.3f} | {avg:.4f} | {r.get('Percentage')} |")
        if KERNEL in name and avg_ms is None:
            avg_ms = avg
            kernel = name.replace("void nbk::", "").replace("(nbk::F32Args, nbk::F32SymShape)", "").replace("(nbk::F32Args)", "")
        if KERNEL in name:
            force_total += float(r.get("TotalDurationNs", 0))
        if REDUCER in name:
            reduce_total += float(r.get("TotalDurationNs", 0))
    # per-dispatch registers from the trace
    for r in rows(f"{out}/stats/**/*kernel_trace.csv"):
        if KERNEL in r.get("Kernel_Name", ""):
            lines += ["", f"force kernel dispatch: grid {r.get('Grid_Size_X')} x wg {r.get('Workgroup_Size_X')}, "
                          f"VGPR {r.get('VGPR_Count')}, SGPR {r.get('SGPR_Count')}, LDS {r.get('LDS_Block_Size')} B, "
                          f"scratch {r.get('Scratch_Size')}"]
            break

    # 2./3./4. counters: one row per dispatch and counter
    def counter(sub, name, kernel=KERNEL):
        vals = [float(r["Counter_Value"]) for r in rows(f"{out}/{sub}/**/*counter_collection.csv")
                if r.get("Counter_Name") == name and kernel in r.get("Kernel_Name", "")]
        return vals

    fetch = counter("pmc_fetch", "FETCH_SIZE")
    write = counter("pmc_write", "WRITE_SIZE")
    rfetch = counter("pmc_fetch", "FETCH_SIZE", REDUCER)
    rwrite = counter("pmc_write", "WRITE_SIZE", REDUCER)
    lines += ["", "## --pmc (separate passes)", ""]
    traffic = force_bytes = reducer_bytes = None
    if fetch and write:
        f_kib, w_kib = sum(fetch) / len(fetch), sum(write) / len(write)
        force_bytes = traffic = (2.0 * f_kib + w_kib) * 1024.0
        lines += [f"* force kernel: FETCH_SIZE = {f_kib:.1f} KiB per launch (raw) -> x2 gfx950 correction = {2 * f_kib * 1024 / 1e6:.2f} MB read",
                  f"* force kernel: WRITE_SIZE = {w_kib:.1f} KiB per launch = {w_kib * 1024 / 1e6:.2f} MB written",
                  f"* force kernel: HBM traffic per launch = {traffic / 1e6:.2f} MB"]
        if rfetch and rwrite:
            rf, rw = sum(rfetch) / len(rfetch), sum(rwrite) / len(rwrite)
            reducer_bytes = (2.0 * rf + rw) * 1024.0
            traffic += reducer_bytes
            lines += [f"* reducer: FETCH_SIZE = {rf:.1f} KiB (raw, x2 = {2 * rf * 1024 / 1e6:.2f} MB read), WRITE_SIZE = {rw:.1f} KiB "
                      f"= {rw * 1024 / 1e6:.2f} MB written -> {reducer_bytes / 1e6:.2f} MB per launch",
                      f"* **HBM traffic per step (force + reducer, the launches kernel_ms spans) = {traffic / 1e6:.2f} MB**"]
    sq = {}
    for name in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY",
                 "SQ_INSTS_LDS", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE"):
        v = counter("pmc_sq", name)
        if v:
            sq[name] = sum(v) / len(v)
            lines.append(f"* {name} = {sq[name]:.4g} per launch")
    if "SQ_ACTIVE_INST_VALU" in sq and "SQ_WAVE_CYCLES" in sq and sq["SQ_WAVE_CYCLES"]:
        lines.append(f"* VALU-active share of wave cycles = {sq['SQ_ACTIVE_INST_VALU'] / sq['SQ_WAVE_CYCLES']:.3f}")
    if "SQ_ACTIVE_INST_VALU" in sq and "SQ_BUSY_CYCLES" in sq and sq["SQ_BUSY_CYCLES"]:
        lines.append(f"* SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES = {sq['SQ_ACTIVE_INST_VALU'] / sq['SQ_BUSY_CYCLES']:.3f}")
    valu_busy = None
    if "SQ_ACTIVE_INST_VALU" in sq and sq.get("GRBM_GUI_ACTIVE"):
        # SQ_ACTIVE_INST_* count quad-cycles summed over the SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
        valu_busy = sq["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * sq["GRBM_GUI_ACTIVE"] / 8)
        lines.append(f"* VALU busy = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) = **{valu_busy:.3f}**")
        if avg_ms:
            lines.append(f"* effective clock = GRBM_GUI_ACTIVE / 8 / kernel time = {sq['GRBM_GUI_ACTIVE'] / 8 / (avg_ms * 1e-3) / 1e9:.3f} GHz")
    if avg_ms:
        lines += ["", f"force kernel average duration (stats pass): **{avg_ms:.3f} ms**"]
    os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
    with open(os.path.join(root, "profiles", f"{tag}_rocprof_summary.md"), "w") as f:
        f.write("\n".join(lines) + "\n")
    # copy the small stats csv verbatim
    for path in glob.glob(f"{out}/stats/**/*kernel_stats.csv", recursive=True):
        with open(path) as src, open(os.path.join(root, "profiles", f"{tag}_kernel_stats.csv"), "w") as dst:
            dst.write(src.read())
    if traffic is not None:
        tpath = os.path.join(root, "profiles", "pmc_traffic.json")
        try:
            t = json.load(open(tpath))
        except Exception:
            t = {}
        key = os.environ.get("NB_TRAFFIC_KEY", "n1048576_p1")
        # bench.py only quotes these numbers for the configuration they were measured on
        t[key] = {"hbm_bytes_per_step": traffic, "force_bytes": force_bytes, "reducer_bytes": reducer_bytes,
                  "fetch_kib_raw": f_kib, "write_kib": w_kib, "tag": tag,
                  "kernel_avg_ms": avg_ms, "kernel": kernel, "j_split": int(os.environ.get("NB_TRAFFIC_JSPLIT", "1" if "sym" in KERNEL else "8")),
                  "reduce_share_of_span": reduce_total / (reduce_total + force_total) if force_total else None,
                  "valu_busy": valu_busy}
        json.dump(t, open(tpath, "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
