"""One line per kernel from a scripts/pmc_passes.sh directory (summary.txt): duration, MFMA-pipe utilisation, and where the
waves' time goes.  Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves,
SQ_VALU_MFMA_BUSY_CYCLES is cycles summed over the 1024 SIMDs, GRBM_GUI_ACTIVE is cycles summed over the 8 XCDs.
  mfma%  = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)          matrix pipe busy
  valu% / lds% / wait% / stall% = SQ_ACTIVE_INST_VALU / _LDS / SQ_WAIT_ANY / SQ_WAIT_INST_ANY over SQ_WAVE_CYCLES
          (share of a wave's life issuing vector-ALU / LDS instructions, parked in s_waitcnt or a barrier, stalled at issue)
  w/simd = 4 x SQ_WAVE_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024)                        resident waves per SIMD, time average
  conf%  = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE                                 LDS cycles lost to bank conflicts
  GHz    = GRBM_GUI_ACTIVE / 8 / duration (reads high on dispatches shorter than ~0.3 ms)
usage: python scripts/pmc_table.py gpurun_out/<dir>/summary.txt"""
import re, sys
rows, cur = {}, None
for line in open(sys.argv[1]):
    if not line.startswith("   "):
        cur = line.strip(); rows[cur] = {}
    else:
        m = re.match(r"\s+(\S+)\s+mean ([\d.e+-]+)", line)
        if m: rows[cur][m.group(1)] = float(m.group(2))
        m = re.match(r"\s+dispatches/pass (\d+)\s+mean duration ([\d.]+) us", line)
        if m: rows[cur]["n"] = int(m.group(1)); rows[cur]["us"] = float(m.group(2))
def short(k): return k.replace("void rpn::", "").replace("rpn::", "").split("(")[0][:58]
print("%-58s %4s %8s %6s %6s %6s %6s %6s %6s %6s %5s %9s %9s" % ("kernel", "n", "us", "mfma%", "valu%", "lds%", "wait%", "stall%", "w/simd", "conf%", "GHz", "fetchMB", "writeMB"))
for k, r in rows.items():
    if "us" not in r or "SQ_WAVE_CYCLES" not in r: continue
    wc = max(r["SQ_WAVE_CYCLES"], 1.0); cyc = max(r.get("GRBM_GUI_ACTIVE", 0) / 8.0, 1.0)
    print("%-58s %4d %8.1f %6.1f %6.1f %6.1f %6.1f %6.1f %6.2f %6.1f %5.2f %9.2f %9.2f" % (
        short(k), r["n"], r["us"], 100 * r.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024),
        100 * r.get("SQ_ACTIVE_INST_VALU", 0) / wc, 100 * r.get("SQ_ACTIVE_INST_LDS", 0) / wc, 100 * r.get("SQ_WAIT_ANY", 0) / wc,
        100 * r.get("SQ_WAIT_INST_ANY", 0) / wc, 4 * wc / (cyc * 1024), 100 * r.get("SQ_LDS_BANK_CONFLICT", 0) / max(r.get("SQ_LDS_IDX_ACTIVE", 1), 1),
        cyc / (r["us"] * 1e3), 2 * 1024 * r.get("FETCH_SIZE", 0) / 1e6, 1024 * r.get("WRITE_SIZE", 0) / 1e6))
