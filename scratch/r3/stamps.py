"""Print the s_memtime stamps a -DWINO_STAMP build of wino.hip left (SPRK_WINO_STAMP file): cycles relative to the workgroup's
earliest stamp 0, iteration 10 of the first tile, per wave.
0 start | 1 DMA issued | 2 transform of the next chunk | 3-6 positions 0-3 issued | 7 vmcnt(0) done (the barrier follows)
A stamp is taken when the wave reaches it in program order: MFMAs still queued behind it count into the next phase."""
import sys
rows = [l.split() for l in open(sys.argv[1])]
for wg in sorted({r[0] for r in rows})[:2]:
    rs = [r for r in rows if r[0] == wg]
    t0 = min(int(r[2]) for r in rs)
    print(wg)
    for r in rs:
        v = [(int(x) - t0) & 0xFFFFFFFF for x in r[2:10]]
        print("  %s " % r[1] + " ".join("%6d" % x for x in v) + "   deltas " + " ".join("%5d" % (v[i + 1] - v[i]) for i in range(7)))

# tile level and clock: 8/9 = memtime/realtime after the first tile's K loop, 10/11 at kernel start, 12 first tile done, 13/14 last tile done
for r in rows[:2] + rows[4:6]:
    v = [int(x) for x in r[2:18]]
    d = lambda a, b: (v[a] - v[b]) & 0xFFFFFFFF
    print(r[0], r[1], "start->K loop end %d cyc | first tile %d cyc | kernel %d cyc = %.1f us (realtime) -> %.3f GHz" % (d(8, 10), d(12, 10), d(13, 10), d(14, 11) / 100.0, d(13, 10) / (d(14, 11) * 10.0)))
