"""VGPR / scratch use of the kernels of libsprk.so: extracts the gfx950 code objects from the .hip_fatbin section
(clang offload bundle) and reads the AMDGPU metadata note.  python scratch/r4/regs.py [name filter]"""
import re, subprocess, sys, os
so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "spr_pick_amd", "libsprk.so")
B = "/opt/rocm/lib/llvm/bin/"
subprocess.check_call([B + "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, "/tmp/fatbin.bin"])
data = open("/tmp/fatbin.bin", "rb").read()
flt = sys.argv[1] if len(sys.argv) > 1 else ""
pos, k = 0, 0
while True:
    i = data.find(b"__CLANG_OFFLOAD_BUNDLE__", pos)
    if i < 0:
        break
    import struct
    n = struct.unpack_from("<Q", data, i + 24)[0]
    off = i + 32
    for _ in range(n):
        o, sz, ln = struct.unpack_from("<QQQ", data, off)
        name = data[off + 24:off + 24 + ln].decode()
        off += 24 + ln
        if "gfx950" in name and sz:
            path = "/tmp/co_%d.co" % k
            open(path, "wb").write(data[i + o:i + o + sz])
            txt = subprocess.run([B + "llvm-readelf", "--notes", path], capture_output=True, text=True).stdout
            for m in re.finditer(r"\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)", txt, re.S):
                if flt in m.group(1):
                    print("%-100s scratch %5s sgpr %4s vgpr %4s" % (m.group(1)[:100], m.group(2), m.group(3), m.group(4)))
            k += 1
    pos = i + 24
