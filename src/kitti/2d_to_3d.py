#!/usr/bin/env python3
"""Drop-in for the reference's src/kitti/2d_to_3d.py (see cm3d_amd/pipeline_kitti.py for the inputs)."""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")))

from cm3d_amd.pipeline_kitti import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main())
