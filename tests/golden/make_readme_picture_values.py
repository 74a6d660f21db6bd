"""What readme_files/MuliPhase1.PNG of the reference shows, as numbers: the picture of a settled two-phase tank.

    python tests/golden/make_readme_picture_values.py [/root/reference]  ->  tests/golden/readme_pictures.json

The picture is the only OUTPUT of the path the reference holds for two phases.  Pixels are classed by colour (the two
SpawnBox.particleColor values of Scene 2 / 4 under Unity's lighting: a yellow and a blue family); the fixture keeps, for
each family, the pixel count, the mean colour and the rows it covers.  tests/test_readme_pictures_gpu.py renders the
engine's settled two-phase tank with sph_amd.statefile.splat_png, classes its pixels the same way and asks for the
same arrangement (yellow band above blue band, each a single band across the tank).  Needs PIL; run by hand here."""
import json
import os
import sys

import numpy as np


def classes(rgb):
    r, g, b = (rgb[..., k].astype(np.int32) for k in range(3))
    yellow = (r > b + 40) & (g > b + 40)
    blue = (b > r + 40) & (b > g + 15)
    return yellow, blue


def describe(rgb):
    """Pixel statistics of the two colour families (rows count from the top of the picture)."""
    out = {}
    yellow, blue = classes(rgb)
    for name, m in (("yellow", yellow), ("blue", blue)):
        ys, xs = np.nonzero(m)
        out[name] = dict(pixels=int(m.sum()), mean_rgb=[round(float(v), 1) for v in rgb[m].mean(axis=0)],
                         mean_row=float(ys.mean()), row_p05=float(np.percentile(ys, 5)), row_p95=float(np.percentile(ys, 95)),
                         col_p05=float(np.percentile(xs, 5)), col_p95=float(np.percentile(xs, 95)))
    ys = np.nonzero(yellow | blue)[0]
    mid = 0.5 * (np.percentile(ys, 2) + np.percentile(ys, 98))          # half way up the fluid
    rows = np.arange(rgb.shape[0])[:, None]
    upper, lower = rows < mid, rows >= mid
    out["yellow_share_of_upper_half"] = float((yellow & upper).sum() / max(((yellow | blue) & upper).sum(), 1))
    out["blue_share_of_lower_half"] = float((blue & lower).sum() / max(((yellow | blue) & lower).sum(), 1))
    out["height"], out["width"] = int(rgb.shape[0]), int(rgb.shape[1])
    return out


def main():
    from PIL import Image
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    here = os.path.dirname(os.path.abspath(__file__))
    rgb = np.asarray(Image.open(os.path.join(ref, "readme_files", "MuliPhase1.PNG")).convert("RGB"))
    doc = {"source": "readme_files/MuliPhase1.PNG of CPearson109/SPH (pixel statistics only)", "MuliPhase1.PNG": describe(rgb)}
    with open(os.path.join(here, "readme_pictures.json"), "w") as fh:
        json.dump(doc, fh, indent=1, sort_keys=True)
        fh.write("\n")
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
