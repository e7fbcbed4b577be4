"""Developer A/B harness (GPU box): run bench.py against an alternative build of the library.

    hipcc ... -DSOME_EXPERIMENT -o build_variants/libB.so          (see tools/build_variant.sh)
    python tools/ab_bench.py build_variants/libB.so --config 1 --steps 3 --no-cpu-baseline --no-extra-configs

The product binding always loads csrc/libwavernn_amd.so; this script points it at another file before the first load.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[2:]
from tacotronv2_wavernn_chinese_amd import _cabi  # noqa: E402

_cabi.LIB_PATH = lib
import bench  # noqa: E402

sys.exit(bench.main())
