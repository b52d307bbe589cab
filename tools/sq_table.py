#!/usr/bin/env python3
"""Join the raw SQ counters of a 2^20 query (tools/gpu_r4_sq_tails.sh: per-kernel sums over one warm-up + one measured query) with the
single-lane kernel table of the same build (tools/kernel_rooflines.py): per kernel the VALU instructions, the time the vector ALUs
alone need at one instruction per SIMD per four cycles, the duration, the bytes and their HBM time, and what that says.
Usage: sq_table.py <sq_counters_tails_raw.txt> <kernel_rooflines_q20.txt> <header note> > table.txt"""
import re
import sys

raw, table, note = sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else ""
rows = []
for line in open(raw):
    if line.startswith('#'):
        continue
    m = re.match(r'(\S.*?)\s+n=\s*(\d+) vgpr=\s*(\d+).*?valu=([\d.]+) lds=([\d.]+) sca=([\d.]+) wait_inst=([\d.]+) wait_any=([\d.]+).*?waves=(\S+) VALU=(\S+) ', line)
    if m:
        rows.append((m.group(1).strip(), int(m.group(2)) // 2, float(m.group(10)) / 2, float(m.group(4)), float(m.group(7)), float(m.group(8))))
single = {}
for line in open(table):
    p = line.split()
    if len(p) >= 5 and p[0].startswith('k_'):
        try:
            single[' '.join(p[:-4])] = (float(p[-3]), float(p[-2]))
        except ValueError:
            pass
ISSUE = 256 * 4 * 2.4e9 / 4  # wave-instructions per second at one VALU instruction per SIMD per four cycles
print("""SQ counters of every kernel of ONE 2^20 indexScenario (64-block batch, two lanes), rocprofv3 --pmc in two passes (tools/gpu_r4_sq_tails.sh).
%s
Per query:
  VALU        wave-level vector instructions issued (SQ_INSTS_VALU)
  issue ms    VALU / (256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles): the time the vector ALUs alone need at one instruction per SIMD per 4 cycles
  ms, GB      the kernel's single-lane duration and ledger bytes per query (HYDIA_LANES=1 kernel table of the same build)
  HBM ms      GB / 6.5 TB/s (about what loop B sustains)
  valu / wait_inst / wait_any   SQ_ACTIVE_INST_VALU, SQ_WAIT_INST_ANY, SQ_WAIT_ANY over SQ_WAVE_CYCLES (per wave; x waves per SIMD = SIMD busy)
kernel                                      n   VALU(G)  issue ms     ms      GB   HBM ms   valu  wait_inst wait_any  reading""" % note)
tv = tm = th = 0
for name, n, V, valu, wi, wa in sorted(rows, key=lambda r: -r[2]):
    if any(s in name for s in ('repack', 'fill_uniform', 'key_pack', 'rocclr')):
        continue
    ms, gb = single.get(name, (0, 0))
    if ms == 0:
        continue
    iss, hbm = V / ISSUE * 1e3, gb / 6.5
    verdict = 'HBM' if hbm > 0.8 * ms else ('VALU' if iss > 0.62 * ms else ('VALU + HBM (additive)' if iss + hbm > 0.85 * ms else 'latency'))
    print("%-42s %3d %8.3f %8.2f %7.2f %7.2f %7.2f   %5.2f %9.2f %8.2f  %s" % (name[:42], n, V / 1e9, iss, ms, gb, hbm, valu, wi, wa, verdict))
    if 'hydia_tensor' not in name:
        tv += iss
        tm += ms
        th += hbm
print("\nEverything outside loop B (per-block tails + loop A): %.1f ms single-lane = %.1f ms at the VALU issue limit + %.1f ms of HBM time at "
      "6.5 TB/s on the bytes moved." % (tm, tv, th))
