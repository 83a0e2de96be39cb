"""Per-kernel matrix-pipe utilisation from the one-pass counter database of tools/probe_pmc_mfma.sh:

    python tools/pmc_mfma_summary.py gpurun_out/pmc_mfma/p_results.db N_STEPS [out.json] > profiles/rNN_vrnn_pmc_mfma.csv

For every kernel: dispatches, total duration (kernel trace of the same pass), the counter sums, and
  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x duration x effective clock)   — share of the chip's matrix-pipe cycles
  clock     = GRBM_GUI_ACTIVE / 8 XCDs / duration (MI355X_MICROARCH.md, DVFS give-back; reads high on dispatches < 0.3 ms)
  parked / stalled / issuing = SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES.
The json holds the figures bench.py prints as roofline.mfma_busy (dominant kernel = pchain_kernel; the weight-gradient GEMM beside it)."""
import csv
import json
import sqlite3
import sys

SIMDS = 4 * 256


def main():
    db, steps = sys.argv[1], float(sys.argv[2])
    con = sqlite3.connect(db)
    dur, ctr = {}, {}
    # one row per dispatch and counter; `start` / `end` are the dispatch's own stamps (the kernel trace of the same pass)
    for k, c, n, v, ns in con.execute("select kernel_name, counter_name, count(*), sum(value), sum(end - start) from counters_collection "
                                      "group by kernel_name, counter_name"):
        ctr.setdefault(k, {})[c] = v
        dur[k] = (n, ns * 1e-9)
    out = csv.writer(sys.stdout)
    cols = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAVES", "GRBM_GUI_ACTIVE"]
    out.writerow(["kernel", "dispatches_per_step", "ms_per_step", "clock_GHz", "mfma_busy", "parked", "issue_stalled", "issuing"] + cols)
    summary = {}
    for k, (n, sec) in sorted(dur.items(), key=lambda kv: -kv[1][1]):
        c = ctr.get(k, {})
        if not c or sec <= 0:
            continue
        clock = c.get("GRBM_GUI_ACTIVE", 0.0) / 8 / sec
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (SIMDS * sec * clock) if clock else 0.0
        wc = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        row = [k[:120], n / steps, sec * 1e3 / steps, clock * 1e-9, busy, c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc,
               c.get("SQ_ACTIVE_INST_ANY", 0) / wc] + [c.get(x, 0.0) / steps for x in cols]  # fmt: skip
        out.writerow([f"{v:.4g}" if isinstance(v, float) else v for v in row])
        for tag, pat in (("pchain", "pchain_kernel"), ("wgrad_gemm", "gemm_group_kernel")):
            if pat in k and tag not in summary:
                summary[tag] = dict(kernel=k[:160], dispatches_per_step=n / steps, ms_per_step=sec * 1e3 / steps, clock_GHz=clock * 1e-9, mfma_busy=busy,
                                    parked=c.get("SQ_WAIT_ANY", 0) / wc, issue_stalled=c.get("SQ_WAIT_INST_ANY", 0) / wc, issuing=c.get("SQ_ACTIVE_INST_ANY", 0) / wc)  # fmt: skip
    if len(sys.argv) > 3:
        with open(sys.argv[3], "w") as f:
            json.dump(summary, f, indent=1)


if __name__ == "__main__":
    main()
