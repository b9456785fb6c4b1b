/* CPU restatement (plain C) of the reference's greedy 2-D NMS.
 *
 * TEST INFRASTRUCTURE ONLY: linked by tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg; never by libsprk.so or spr_pick_amd/.
 *
 * Follows /root/reference/spr_pick/utils/algorithms.py:59-103 (see oracle/nms.py
 * for the prose).  The reference's Python `set` of suppressed flat indices is a
 * byte map here, sized (H+1)*W + W + 1 so that the out-of-range indices the
 * reference's clip-to-H / clip-to-W produces are representable and harmless,
 * exactly as inserting them into the set is.
 * Tie order: score descending, then flat index descending.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float s; int64_t i; } item_t;

static int cmp_desc(const void *pa, const void *pb) {
    const item_t *a = (const item_t *)pa, *b = (const item_t *)pb;
    if (a->s > b->s) return -1;
    if (a->s < b->s) return 1;
    if (a->i > b->i) return -1;
    if (a->i < b->i) return 1;
    return 0;
}

long oracle_nms2d(const float *x, int H, int W, int r, float threshold,
                  float *out_scores, int32_t *out_xy, long cap) {
    const int64_t n = (int64_t)H * W;
    int64_t m = 0;
    item_t *items = (item_t *)malloc(sizeof(item_t) * (size_t)(n > 0 ? n : 1));
    if (!items) return -1;
    /* only scores > threshold can ever be visited (the walk stops at the first <=) */
    for (int64_t i = 0; i < n; ++i)
        if (x[i] > threshold) { items[m].s = x[i]; items[m].i = i; ++m; }
    qsort(items, (size_t)m, sizeof(item_t), cmp_desc);

    const size_t map_n = (size_t)(H + 1) * W + W + 1;
    uint8_t *S = (uint8_t *)calloc(map_n, 1);
    int *dmax = (int *)malloc(sizeof(int) * (size_t)(2 * r + 1));
    if (!S || !dmax) { free(items); free(S); free(dmax); return -1; }
    for (int di = -r; di <= r; ++di) {
        int d = 0;
        while ((d + 1) * (d + 1) + di * di <= r * r) ++d;
        dmax[di + r] = d;
    }
    long cnt = 0;
    for (int64_t k = 0; k < m; ++k) {
        const int64_t i = items[k].i;
        if (S[i]) continue;
        const int xx = (int)(i % W), yy = (int)(i / W);
        if (cnt >= cap) { cnt = -2; break; }
        out_scores[cnt] = items[k].s;
        out_xy[2 * cnt] = xx;
        out_xy[2 * cnt + 1] = yy;
        ++cnt;
        for (int di = -r; di <= r; ++di) {
            int yc = yy + di;
            yc = yc < 0 ? 0 : (yc > H ? H : yc);
            const int d = dmax[di + r];
            for (int dj = -d; dj <= d; ++dj) {
                int xc = xx + dj;
                xc = xc < 0 ? 0 : (xc > W ? W : xc);
                S[(size_t)yc * W + xc] = 1;
            }
        }
    }
    free(items); free(S); free(dmax);
    return cnt;
}
