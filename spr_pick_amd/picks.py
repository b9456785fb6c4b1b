"""Pick post-filter and ``*_scores.txt`` writer — the consumer of the NMS output in the reference
(train.py:557-571): keep picks farther than 30 px from every border of the un-padded score map
and write ``name<TAB>coords[i,1]<TAB>coords[i,0]<TAB>score`` under the header
``image_name x_coord y_coord score``.  (The reference's tensors are H/W-transposed, so array
row = real x: column order (row, col) is what lands in the x/y columns — reproduced as is.)"""
import numpy as np

BORDER = 30


def filter_picks(scores, coords, shape, border=BORDER):
    """coords[:,0] = column (xx), coords[:,1] = row (yy) as returned by non_maximum_suppression;
    shape = (rows, cols) of the score map."""
    scores = np.asarray(scores)
    coords = np.asarray(coords).reshape(-1, 2)
    x_max, y_max = shape[0] - border, shape[1] - border
    keep = (coords[:, 1] > border) & (coords[:, 1] < x_max) & (coords[:, 0] > border) & (coords[:, 0] < y_max)
    return scores[keep], coords[keep]


def write_scores(path, name, scores, coords, shape, border=BORDER):
    s, c = filter_picks(scores, coords, shape, border)
    with open(path, "w") as f:
        print("image_name\tx_coord\ty_coord\tscore", file=f)
        for i in range(len(s)):
            print(name + "\t" + str(c[i, 1]) + "\t" + str(c[i, 0]) + "\t" + str(s[i]), file=f)
    return len(s)
