import os, sys, time, subprocess
# timing-only ablations of the conv kernel: results are WRONG by construction, only durations matter
code = '''
import os, sys, time, torch, ctypes as C
sys.path.insert(0, %r)
import popnet_amd
from popnet_amd.pipeline import PoseEngine
from popnet_amd import synth, _lib
eng = PoseEngine(precision="bf16", device="cuda:0", max_batch=32)
if os.environ.get("POPNET_ZERO_W"):
    with torch.no_grad():
        for p_ in eng.model.parameters(): p_.zero_()
    eng.model.invalidate()     # the engine re-reads the handle (PoseEngine.net)
d = torch.from_numpy(synth.synth_depth(32)).cuda()
B = eng.preprocess(d)
for _ in range(5): eng.forward(B)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): eng.forward(B)
torch.cuda.synchronize(); print("DBG", os.environ.get("POPNET_DBG"), "forward ms %%.3f" %% ((time.perf_counter()-t0)/30*1e3))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for dbg in ("0", "8", "4", "Z"):
    env = dict(os.environ, POPNET_DBG=dbg if dbg != "Z" else "0")
    if dbg == "Z": env["POPNET_ZERO_W"] = "1"
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:])
