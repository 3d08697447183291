"""profiles/<round>/mfma_busy.txt from the PMC + trace tables of a profile directory:
    python tools/mfma_busy.py profiles/r04 > profiles/r04/mfma_busy.txt
MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES per dispatch / (avg us of the kernel trace x 2400 cycles/us x 1024 SIMDs), i.e. at the NOMINAL
clock (a kernel that holds the chip below 2.4 GHz shows less than its pipe occupancy); wait share = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES.
"""
import os
import sys


def table(path):
    rows = []
    for line in open(path):
        cells = [c.strip() for c in line.strip().strip("|").split("|")]
        if len(cells) >= 5 and cells[0] not in ("kernel", "---") and not set(cells[0]) <= set("-"):
            rows.append(cells)
    return rows


def section(d, suffix, title):
    stats = os.path.join(d, "kernel_stats%s.md" % suffix)
    pmc = os.path.join(d, "pmc_sq%s.md" % suffix)
    if not (os.path.exists(stats) and os.path.exists(pmc)):
        return
    avg = {}
    order = []
    for r in table(stats):
        try:
            avg[r[0]] = (int(r[1]), float(r[3]))
            order.append(r[0])
        except ValueError:
            pass
    cnt = {}
    for r in table(pmc):
        try:
            cnt.setdefault(r[0], {})[r[1]] = float(r[4])
        except ValueError:
            pass
    print("## %s  (SQ_VALU_MFMA_BUSY_CYCLES per dispatch / (avg us x 2400 cycles/us x 1024 SIMDs); SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES)" % title)
    for k in order:
        c = cnt.get(k)
        if not c or "SQ_VALU_MFMA_BUSY_CYCLES" not in c or c["SQ_VALU_MFMA_BUSY_CYCLES"] <= 0:
            continue
        n, us = avg[k]
        busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (us * 2400.0 * 1024.0)
        wait = c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else float("nan")
        line = "%-36s launches %5d avg %7.2f us  MFMA-busy %.3f  wait_inst/wave_cycles %.2f" % (k, n, us, busy, wait)
        # GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / the dispatch duration under the SAME pass (pmc_per_kernel.py's DURATION_NS
        # rows).  Round 5 finding: for launches of 7-70 us this ratio comes out ABOVE the 2.4 GHz the chip can run at (the counter
        # window is wider than the dispatch's start..end timestamps), so it is printed as a window ratio, not used as a clock;
        # the clock under load is measured by the probe kernel (tools/effective_clock.py -> effective_clock.md).
        if c.get("GRBM_GUI_ACTIVE") and c.get("DURATION_NS"):
            line += "  | profiled pass: %.2f us, GRBM_GUI_ACTIVE / 8 XCDs / duration = %.2f cycles/ns (window ratio, not a clock)" % (
                c["DURATION_NS"] / 1e3, c["GRBM_GUI_ACTIVE"] / 8.0 / c["DURATION_NS"])
        print(line)


if __name__ == "__main__":
    d = sys.argv[1]
    section(d, "", "f32 headline")
    section(d, "_x3", "package default")
