"""Input fixtures for the inlet-image tests: two of the reference's inlet images (INPUT DATA of the reference,
NavierStokes/InletImages/*.png), box-filtered down so that they stay small, stored as 8-bit grayscale PNGs under
tests/golden/.  Run once in a container that has /root/reference (the GPU box does not); the tests only read the
committed copies."""
import os

import numpy as np
from PIL import Image

SRC = "/root/reference/NavierStokes/InletImages"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main():
    import sys
    sys.path.insert(0, os.path.dirname(OUT.rstrip("/")).rsplit("/tests", 1)[0])
    from stabilized_navier_stokes_flow_fenicsx_amd.inlet_contours import load_image
    for name, f in (("PlusF_final", 4), ("asym_offset", 1), ("Triangle", 2)):
        g = load_image(os.path.join(SRC, name + ".png"))
        h, w = (g.shape[0] // f) * f, (g.shape[1] // f) * f
        g = g[:h, :w].reshape(h // f, f, w // f, f).mean(axis=(1, 3))
        Image.fromarray(np.round(g * 255).astype(np.uint8), "L").save(os.path.join(OUT, f"inlet_{name}.png"), optimize=True)
        print(name, g.shape, os.path.getsize(os.path.join(OUT, f"inlet_{name}.png")), "bytes")


if __name__ == "__main__":
    main()
