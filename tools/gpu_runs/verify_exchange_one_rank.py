import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/realism-effects_amd")
import importlib.util
spec = importlib.util.spec_from_file_location("bench", "/root/repo/bench.py"); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
from rfx_amd.context import Context
ctx = Context(256, 144)
ctx.comm_init(Context.comm_unique_id(), 0, 1)
print("verify_exchange (1-rank ring):", b.verify_exchange(ctx, 0, 1))
ctx.close()
