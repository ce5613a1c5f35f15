"""Where the device's libm_f32.h parts from the C library: mismatch counts of rgbd360_selftest_libm per window (GPU box).
python tools/libm_dev_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgbd360_amd.register import RegisterPhotoICP
reg = RegisterPhotoICP()
for name, first, count in (("0.25-0.5", 0x3e800000, 1 << 23), ("0.5-0.975", 0x3f000000, 0x3f79999a - 0x3f000000), ("0.975-1", 0x3f79999a, 0x3f800000 - 0x3f79999a + 1),
                           ("2^-27..2^-20", 0x32000000, 1 << 23), ("-0.5..-1", 0xbf000000, 1 << 23), ("1e3..", 0x44000000, 1 << 23), ("2..4", 0x40000000, 1 << 23)):
    print(name, hex(first), count, reg.selftest_libm(first, count), flush=True)
