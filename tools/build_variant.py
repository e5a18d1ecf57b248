"""A/B builds: python tools/build_variant.py NAME [-DFOO=1 ...] -> prifit_amd/lib/variants/NAME.so (own object directory).
On the GPU box: PRIFIT_LIB=$PWD/prifit_amd/lib/variants/NAME.so python bench.py ... (tools/ab_libs.sh alternates two of them; the
product library is never overwritten)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
name, defs = sys.argv[1], sys.argv[2:]
os.environ["PRIFIT_BUILD_DEFS"] = " ".join(defs)
from prifit_amd import build
build.OBJDIR = os.path.join(build.HERE, "build_" + name)
vdir = os.path.join(build.LIBDIR, "variants")
os.makedirs(vdir, exist_ok=True)
build.LIB = os.path.join(vdir, name + ".so")
build.COMMON = [c for c in build.COMMON] + [d for d in defs if d not in build.COMMON]
print(build.build_library(verbose=False))
