#!/usr/bin/env python3
"""Condense PMC summaries (scripts/pmc_summary.py output) into profiles/pmc_counters.json, the file
bench.py reads for `roofline.traffic` and `roofline.counter_check`.

    python scripts/pmc_to_json.py c3=profiles/r2_c3_pmc_<tag>.txt c2=... c4=... c5=...

Per workload: memory-side bytes per render_kernel launch (reads = TCC_EA0_RDREQ_sum * 64 B * 2: gfx950
tallies 128-byte read requests at 64 B; writes = WRREQ_64B * 64 + (WRREQ - WRREQ_64B) * 32), the f64
wave-instruction counts, the average number of active lanes per VALU instruction and the VALU busy
fraction (SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * GRBM_GUI_ACTIVE / 8))."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse(path):
    c = {}
    for line in open(path):
        if line.startswith("#") or not line.strip():
            continue
        parts = line.split()
        c[parts[0]] = float(parts[1])
    return c


def main():
    out_path = os.path.join(ROOT, "profiles", "pmc_counters.json")
    try:
        out = json.load(open(out_path))
    except (OSError, ValueError):
        out = {}
    out["_comment"] = __doc__.split("Per workload:")[1].strip().replace("\n", " ")
    for arg in sys.argv[1:]:
        workload, path = arg.split("=", 1)
        c = parse(path)
        rd, wr, wr64 = c["TCC_EA0_RDREQ_sum"], c["TCC_EA0_WRREQ_sum"], c["TCC_EA0_WRREQ_64B_sum"]
        lanes = c["SQ_THREAD_CYCLES_VALU"] / c["SQ_INSTS_VALU"]
        cycles = c["GRBM_GUI_ACTIVE"] / 8.0
        out[workload] = {
            "n_gpus": 1,
            "read_bytes": int(rd * 64 * 2),
            "write_bytes": int(wr64 * 64 + (wr - wr64) * 32),
            "valu_wave_insts": int(c["SQ_INSTS_VALU"]),
            "f64_add_wave_insts": int(c["SQ_INSTS_VALU_ADD_F64"]),
            "f64_mul_wave_insts": int(c["SQ_INSTS_VALU_MUL_F64"]),
            "f64_fma_wave_insts": int(c["SQ_INSTS_VALU_FMA_F64"]),
            "active_lanes_per_valu_inst": lanes,
            "kernel_cycles": cycles,
            "valu_busy": c["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cycles),
            "source": os.path.relpath(path, ROOT),
        }
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
