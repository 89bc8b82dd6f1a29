#!/usr/bin/env python3
"""corun_loop.inc for tools/ubench/corun.hip: one 'stage' of a register-only stand-in for K2's plane loop -- 768 v_bitop3 on 64
accumulators (v0..v63) with 8 + 8 operand registers placed so that no instruction has three sources of equal index parity
(half rate otherwise), followed by the 96 half-rate popcount / pack instructions of count_group."""
import sys
out = []
for k in range(12):
    for c in range(8):
        for r in range(8):
            out.append("v_bitop3_b32 v%d, v%d, v%d, v%d bitop3:0xf6" % (8 * r + c, 8 * r + c, 64 + r, 72 + (c ^ 1)))
for r in range(8):
    for c2 in range(4):
        out.append("v_bcnt_u32_b32 v%d, v%d, v%d" % (80 + 4 * r + c2, 8 * r + 2 * c2, 80 + 4 * r + c2))
        out.append("v_bcnt_u32_b32 v112, v%d, 0" % (8 * r + 2 * c2 + 1))
        out.append("v_lshl_add_u32 v%d, v112, 16, v%d" % (80 + 4 * r + c2, 80 + 4 * r + c2))
with open(sys.argv[1], "w") as f:
    for l in out:
        f.write('"%s\\n\\t"\n' % l)
