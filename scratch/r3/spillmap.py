"""scratch/r3/spillmap.py <asm> <symbol pattern>: scratch loads / stores, MFMAs and register moves per basic block of one kernel."""
import re, sys
txt = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
start = next(i for i, l in enumerate(txt) if re.match(r"^_ZN\S*" + pat + r"\S*:", l))
end = next(i for i in range(start, len(txt)) if ".Lfunc_end" in txt[i])
blk = 'entry'; stats = {}; order = []
for l in txt[start:end]:
    m = re.match(r'^(\.LBB\S+):', l)
    if m: blk = m.group(1)
    if blk not in stats: stats[blk] = [0] * 6; order.append(blk)
    t = l.strip().split(' ')[0] if l.strip() else ''
    for i, p in enumerate(('scratch_load', 'scratch_store', 'v_mfma', 'v_mov', 'v_readlane', 'v_writelane')):
        if t.startswith(p): stats[blk][i] += 1
for b in order:
    s = stats[b]
    if s[0] + s[1] > 0 or s[2] > 0: print(b, 'sload', s[0], 'sstore', s[1], 'mfma', s[2], 'mov', s[3], 'readlane', s[4], 'writelane', s[5])
