"""Development builds of the library next to the product: lib/libicp_hip_<name>.so with extra -D defines (then ICP_HIP_LIB=<path>).
usage: python tools/dev_build_variant.py <name> [DEFINE=VALUE ...]   e.g.  times ICP_DEBUG_STEPS=1 ICP_DEBUG_TIMES=1"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import __graft_entry__ as g
name = sys.argv[1]; defs = tuple(sys.argv[2:])
g.build_hip(force=True, variant=name, defines=defs)
