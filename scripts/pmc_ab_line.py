#!/usr/bin/env python3
"""One line of derived figures from the counter CSVs of one build (scripts/pmc_ab.sh): last dispatch of the frame kernel."""
import csv, glob, os, sys

d, name = sys.argv[1], sys.argv[2]
flt = sys.argv[3] if len(sys.argv) > 3 else "render_kernel<"
res = {}
for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if flt not in row["Kernel_Name"] or "weighted" in row["Kernel_Name"]:
                continue
            res.setdefault(row["Counter_Name"], {})[int(row["Dispatch_Id"])] = float(row["Counter_Value"])
c = {k: v[max(v)] for k, v in res.items()}
g = lambda k: c.get(k, float("nan"))
cycles = g("GRBM_GUI_ACTIVE") / 8
valu = g("SQ_INSTS_VALU")
f64 = g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_FMA_F64")
print(f"{name:28s} cycles {cycles / 1e6:7.3f} M  VALU {valu / 1e6:7.1f} M  SALU {g('SQ_INSTS_SALU') / 1e6:6.1f} M  SMEM {g('SQ_INSTS_SMEM') / 1e6:5.1f} M  "
      f"VMEM {g('SQ_INSTS_VMEM') / 1e6:5.1f} M  LDS {g('SQ_INSTS_LDS') / 1e6:5.1f} M  branch {g('SQ_INSTS_BRANCH') / 1e6:5.1f} M  "
      f"lanes {g('SQ_THREAD_CYCLES_VALU') / valu:5.1f}  VALU busy {g('SQ_ACTIVE_INST_VALU') * 4 / (1024 * cycles):5.3f}  "
      f"f64 share {f64 / valu:5.3f}  trans_f64 {g('SQ_INSTS_VALU_TRANS_F64') / 1e6:5.1f} M  "
      f"wait_inst/wave_cycles {g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'):5.3f}  sca_busy/valu_busy {g('SQ_ACTIVE_INST_SCA') / g('SQ_ACTIVE_INST_VALU'):5.3f}")
