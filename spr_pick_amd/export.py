"""Pick exporters: STAR tables and the score-file -> STAR conversion.

* ``write_star`` / ``coordinates_to_star``: utils/star.py:101-108 and utils/conversions.py:73-91 —
  ``data_images`` / ``loop_`` header, one ``_rln<Name> #k`` line per column in table order, then
  tab-separated rows; coordinate-table columns renamed to RELION's and the micrograph extension
  appended.
* ``scores_to_star``: the repository-root ``convert_to_star.py`` with its constants as arguments:
  every ``*_scores.txt`` row with score > threshold and strictly inside the (x, y) window is
  written as ``int(x*scale) TAB int(y*scale) TAB <micrograph>.mrc TAB score`` under the
  ``# version 30001`` header.  The script strips a fixed 18-character suffix from the file name to
  get the micrograph name; ``strip`` is that constant (use ``len("_scores.txt")`` for the
  evaluator's ``{name}_scores.txt`` files)."""
import os

RELION_NAMES = {"score": "AutopickFigureOfMerit", "image_name": "MicrographName", "x_coord": "CoordinateX",
                "y_coord": "CoordinateY", "voltage": "Voltage", "detector_pixel_size": "DetectorPixelSize",
                "magnification": "Magnification", "amplitude_contrast": "AmplitudeContrast"}

STAR_HEADER = ("# version 30001\n\ndata_\n\nloop_\n_rlnCoordinateX #1\n_rlnCoordinateY #2\n"
               "_rlnMicrographName #3\n_rlnAutopickFigureOfMerit #4\n")


def coordinates_to_star(table, image_ext=""):
    table = table.copy()
    for ours, relion in RELION_NAMES.items():
        if ours in table.columns:
            table[relion] = table[ours]
            table = table.drop(ours, axis=1)
    table["MicrographName"] = table["MicrographName"].apply(lambda n: n + image_ext)
    return table


def write_star(table, f):
    print("data_images", file=f)
    print("loop_", file=f)
    for i, name in enumerate(table.columns):
        print("_rln" + name + " #" + str(i + 1), file=f)
    table.to_csv(f, sep="\t", index=False, header=False)


def scores_to_star(score_files, out_path, threshold=0.13, x_range=(15, 1425), y_range=(15, 1009), scale=4,
                   strip=18, image_ext=".mrc"):
    """-> number of particles written."""
    import pandas as pd
    n = 0
    with open(out_path, "w") as f:
        f.write(STAR_HEADER)
        for path in score_files:
            name = os.path.basename(path)[:-strip] + image_ext
            rows = pd.read_csv(path, sep="\t")
            for x, y, s in zip(rows.x_coord, rows.y_coord, rows.score):
                if s > threshold and x_range[0] < x < x_range[1] and y_range[0] < y < y_range[1]:
                    f.write(str(int(x * scale)) + "\t" + str(int(y * scale)) + "\t" + name + "\t" + str(s) + "\n")
                    n += 1
    return n
