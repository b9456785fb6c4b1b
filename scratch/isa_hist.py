"""Histogram the basic blocks that contain MFMAs for one kernel symbol pattern in /tmp/conv.s."""
import re, collections, sys
txt = open('/tmp/conv.s').read().split("\n")
pat = sys.argv[1]
start = next(i for i, l in enumerate(txt) if re.match(r"^_ZN\S*" + pat + r"\S*:", l))
end = next(i for i in range(start, len(txt)) if ".Lfunc_end" in txt[i])
lines = txt[start:end]
print("lines", len(lines))
blocks = []; cur = ("entry", [])
for l in lines:
    if re.match(r"^\.LBB\S+:", l):
        blocks.append(cur); cur = (l.strip(), [])
    elif l.strip() and not l.strip().startswith((";", ".")):
        cur[1].append(l.strip())
blocks.append(cur)
for name, ins in blocks:
    ops = [i.split()[0] for i in ins]
    c = collections.Counter(ops)
    nm = sum(v for k, v in c.items() if k.startswith("v_mfma"))
    if nm >= 8:
        ov = sum(v for k, v in c.items() if k.startswith("v_") and not k.startswith("v_mfma"))
        print(name, "n=%d mfma=%d valu_other=%d salu=%d ds=%d" % (len(ins), nm, ov, sum(v for k, v in c.items() if k.startswith("s_")), sum(v for k, v in c.items() if k.startswith("ds_"))))
        print("   ", c.most_common(16))
        if len(sys.argv) > 2 and sys.argv[2] == name.rstrip(":"):
            print("\n".join(ins))
