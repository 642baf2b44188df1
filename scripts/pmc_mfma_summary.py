"""Summarise a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE pass.

Per kernel (template arguments kept, parameter list dropped):
  mfma_flops  = SQ_INSTS_VALU_MFMA_MOPS_F64 * 512            (rocprofv3's own MfmaFlopsF64 expression)
  mfma_util   = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)
rocprofv3 reports one value per dispatch with the counter's instances summed: SQ counters over all SIMDs, GRBM_GUI_ACTIVE
over the 8 XCDs -- hence the / 8 (check: gui_active/8 divided by the traced kernel time gives the shader clock, ~2.4 GHz)
and the 1024 SIMDs (256 CUs x 4).  ROCm 7.2 has no gfx950 section for the derived MfmaUtil (MI355X_MICROARCH.md), so the
ratio is formed here from the raw counters."""
import csv, sys, collections

agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter(); ns = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "gemm_f64_kernel" in name and int(r["Grid_Size"]) >= (1 << 23):
        name += "[grid >= 2^23 threads: the H = V0 E V1^T contraction]"
    agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        n[name] += 1
        ns[name] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
print("kernel,launches,traced_ms,mfma_flops,mfma_TFLOPs_over_traced_time,mfma_busy_cycles,gui_active_per_xcd,clock_GHz,mfma_util_pct,sq_busy_cycles")
for name, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0)):
    gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    if busy <= 0:
        continue
    fl = c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0) * 512
    ms = ns[name] / 1e6
    print(f"{name},{n[name]},{ms:.3f},{fl:.4g},{fl / (ms * 1e-3) / 1e12:.2f},{busy:.4g},{gui:.4g},{gui / (ms * 1e6):.2f},"
          f"{100 * busy / (gui * 1024):.1f},{c.get('SQ_BUSY_CYCLES', 0.0):.4g}")
