"""CPU-side probe of the pipe schedule (diagnostic): builds the schedule of selected subdomains of an N^3 / 2x2x2 problem with the
test harness and dumps per task: group sweep W nsteps active-rows nprod wide-steps start-level (+ producers and, in
gpurun_out/pipe_needs.bin, the steps required of each producer at every step).  usage: python tools/pipe_schedule_probe.py N SUBDOMAINS(e.g. 0,7)"""
import os, sys, ctypes, numpy as np, scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
ddm = ge.import_package()
from dune_ddm_amd import synth
from dune_ddm_amd.problem import build_structured
import test_pipe_schedule as tps
import subprocess
subprocess.check_call(["make", "-C", tps.CPP, "libpipe_host_test.so"], stdout=subprocess.DEVNULL)
lib = ctypes.CDLL(os.path.join(tps.CPP, "libpipe_host_test.so")); lib.pipe_test_build_and_emulate.restype = ctypes.c_int
N = int(sys.argv[1])
dec = build_structured(synth.StructuredPoisson((N, N, N), (2, 2, 2)), overlap=2, pou_type="distance", shrink=0)
which = [int(a) for a in sys.argv[2].split(",")]
mats = [dec.subs[k].A_dir.tocsr() for k in which]
M = sp.block_diag(mats, format="csr"); M.sort_indices()
bp = np.concatenate([[0], np.cumsum([m.shape[0] for m in mats])]).astype(np.int64)
OUT = os.environ.get("PIPE_PROBE_OUT", "gpurun_out/pipe")
os.environ["PIPE_DEBUG_TASKS"] = OUT + "_tasks.txt"; os.environ["PIPE_DEBUG_NEEDS"]="1"; os.environ["PIPE_DEBUG_NEEDS_BIN"]=OUT + "_needs.bin"
d = np.random.default_rng(0).standard_normal(M.shape[0])
rc, err, x, xo, st = tps.run_pipe(lib, M, bp, d, delta=int(os.environ.get("PIPE_TEST_DELTA", "24")), vote=int(os.environ.get("PIPE_TEST_VOTE", "1")))
print(rc, err, st, "bitexact", np.array_equal(x, xo))
T = np.array([[int(v) for v in l.split()[:9]] for l in open(OUT + "_tasks.txt")], dtype=np.int64)
for g in sorted(set(T[:, 0])):
    for sw in (0, 1):
        m = (T[:, 0] == g) & (T[:, 1] == sw)
        print(f"group {g} sweep {sw}: tasks {m.sum()} steps {T[m,3].sum()} wide steps {T[m,6].sum()} ({T[m,6].sum()/T[m,3].sum():.1%}) tasks with any wide step {(T[m,6]>0).sum()}  fill {T[m,4].sum()/T[m,3].sum()/64:.2f}  slots with a gathered operand per step {T[m,8].sum()/T[m,3].sum():.2f} of 14")
np.save(OUT + "_tasks.npy", T)
