#!/usr/bin/env python3
"""Golden fixtures of the germline_dir=not_available mode (EE:472-506) from the REFERENCE ITSELF.

Run in the build container only (needs /root/reference and oracle/_ref, i.e. `make -C oracle`):
    python tests/golden/make_golden_default.py

oracle/_ref/ee_ref_driver --default calls the reference's own storeReference, storeDuplicates and
generateFinalOutput_default (EE:2948-3043, compiled from /root/reference where it lies) on the committed panels
tests/golden/{mini_edge,toy_subset} for several default_error values.  main()'s own handling of the argument
(EE:353-363: atof; a value <= 0 becomes 0.01) is not part of those functions: the value handed over is the one main()
would hand over, and the file for "<= 0" is therefore the 0.01 one.  Outputs are data only.
"""
import os
import shutil
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
DRV = os.path.join(ROOT, "oracle", "_ref", "ee_ref_driver")
# tag -> the float main() passes on (EE:353-363)
VALUES = {"0.0120": "0.012", "0.0100": "0.01", "0.00049": "0.00049", "0.123456": "0.123456", "7": "7"}


def main():
    assert os.path.exists(DRV), "run `make -C oracle` first"
    for panel in ("mini_edge", "toy_subset"):
        d = os.path.join(HERE, panel)
        for tag, val in VALUES.items():
            if panel == "toy_subset" and tag != "0.0120":
                continue  # one value on the larger panel (overlapping amplicons -> duplicate = YES rows, CRLF BED) is enough
            out = tempfile.mkdtemp(prefix="ampli_golden_default_")
            r = subprocess.run([DRV, "--default", os.path.join(d, "panel.bed"), os.path.join(d, "refbases.txt"), os.path.join(d, "dups.txt"), val, out],
                               capture_output=True, text=True)
            assert r.returncode == 0, r.stdout + r.stderr
            shutil.copy(os.path.join(out, "positionSpecificNoise_default.txt"), os.path.join(d, f"expected_positionSpecificNoise_default_{tag}.txt"))
            shutil.rmtree(out)
        print(panel, "default tables:", ", ".join(VALUES))


if __name__ == "__main__":
    main()
