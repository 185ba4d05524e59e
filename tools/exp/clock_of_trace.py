"""Effective shader clock of the last traced k_conv2h launch: DBFR_CONV2_TRACE=<file> makes workgroup 0 record
(s_memtime, s_memrealtime) at its start and end; s_memrealtime ticks at 100 MHz."""
import struct, sys
d = open(sys.argv[1], "rb").read(32)
c0, r0, c1, r1 = struct.unpack("<4Q", d)
ns = (r1 - r0) * 10.0
print(f"{sys.argv[1]}: {c1 - c0} shader cycles in {ns / 1e3:.1f} us -> {(c1 - c0) / ns:.3f} GHz")
