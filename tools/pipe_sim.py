"""Discrete-event model of the pipe triangular-solve engine on ONE subdomain (= one XCD): replays the schedule dumped by
tools/pipe_schedule_probe.py (per task: producers; per step: steps required of each producer, the step's W) with a latency model of the
device kernel (csrc/trsv_pipe.hpp) and prints the span of each sweep.  Diagnostic: calibrated against the stamped traces under
profiles/ (r02_pipe_trace_216_nc1.txt: forward 1.57 ms, backward 1.70 ms on subdomain 0), used to rank design changes on the CPU
before they cost GPU time.

Model of a compute wave (times in us):
  * an unblocked step takes STEP (+ WIDE1 / WIDE2 when its widest row needs one / two extra halves);
  * a step's results are in L2 DATA after its end, its progress word is published PUB after its end (drain at the top of the
    next step, then the store) and seen by a poll that is ISSUED later than that (+ L2 travel TRAVEL);
  * the requirements of step t+1 are checked at the top of step t with the poll issued at the end of step t-1; if they hold the
    gathers are prefetched and step t+1 runs unblocked.  Otherwise the check is deferred behind step t: polls every POLL_RT until
    the producers have published, then gathers, then the step: it ends BLOCKED_TAIL after the successful poll returned;
  * 64 workgroups per XCD: a task is dequeued when a slot is free, in queue order; the backward sweep starts when the forward
    sweep of the subdomain is complete.
Variants: --spec = speculative gathers validated against a reset pattern (the data itself, not the progress word, gates a
deferred step); --step, --hop etc. override parameters."""
import argparse
import heapq
import struct
import sys

import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("prefix", help="e.g. gpurun_out/sim/needs_216_0.bin")
ap.add_argument("--tasks", default=None, help="tasks txt (group sweep W nsteps ...); default: derived from the prefix")
ap.add_argument("--step", type=float, default=1.05)
ap.add_argument("--wide1", type=float, default=0.25)
ap.add_argument("--wide2", type=float, default=0.45)
ap.add_argument("--data", type=float, default=0.30, help="result store -> visible in L2")
ap.add_argument("--pub", type=float, default=0.45, help="end of a step -> its progress word visible in L2 (drain + store)")
ap.add_argument("--travel", type=float, default=0.15, help="load issue -> sampled in L2")
ap.add_argument("--rt", type=float, default=0.32, help="L2 round trip of a poll / gather")
ap.add_argument("--fetch", type=float, default=0.55, help="top of a step -> gathers of the next step issued")
ap.add_argument("--tail", type=float, default=0.50, help="operands complete -> end of the step (entries, products, sum, ring write)")
ap.add_argument("--slots", type=int, default=64)
ap.add_argument("--spec", action="store_true")
ap.add_argument("--pair", action="store_true", help="adjacent queue tasks share a workgroup: hand-overs between them through LDS")
ap.add_argument("--quiet", action="store_true")
args = ap.parse_args()

data = open(args.prefix, "rb").read()
tasks_txt = args.tasks or args.prefix.replace("needs_", "tasks_").replace(".bin", ".txt")
meta = np.array([[int(v) for v in ln.split()[:9]] for ln in open(tasks_txt)], dtype=np.int64)
nt = len(meta)
off = 0
prods, needs, stepw = [], [], []
for i in range(nt):
    npd, ns = struct.unpack_from("ii", data, off)
    off += 8
    pr = np.frombuffer(data, dtype=np.int32, count=npd, offset=off)
    off += 4 * npd
    a = np.frombuffer(data, dtype=np.uint16, count=(npd + 2) * ns, offset=off).reshape(ns, npd + 2)
    off += 2 * (npd + 2) * ns
    prods.append(pr.astype(np.int64))
    needs.append(a[:, :npd].astype(np.int64))
    stepw.append(a[:, npd].astype(np.int64))
assert off == len(data)
sweep = meta[:, 1]
nsteps = meta[:, 3]


def step_cost(w):
    return args.step + (args.wide1 if w > 14 else 0.0) + (args.wide2 - args.wide1 if w > 20 else 0.0)


def simulate(sw, t_start):
    ids = [i for i in range(nt) if sweep[i] == sw]
    end = {}          # task -> array of step end times
    first = {}
    free = [t_start] * args.slots
    heapq.heapify(free)
    nblocked = 0
    ntot = 0
    tblocked = 0.0
    for i in ids:     # queue order
        t0 = heapq.heappop(free)
        ns = int(nsteps[i])
        E = np.zeros(ns)
        pr, nd, W = prods[i], needs[i], stepw[i]
        # ready times per step: when the producers' requirement is PUBLISHED / when the DATA is in L2
        if len(pr):
            pubt = np.full(ns, -1e30)
            datt = np.full(ns, -1e30)
            for j, p in enumerate(pr):
                r = nd[:, j]
                ep = end[int(p)]
                m = r > 0
                idx = np.clip(r - 1, 0, len(ep) - 1)
                pubt = np.maximum(pubt, np.where(m, ep[idx] + args.pub, -1e30))
                datt = np.maximum(datt, np.where(m, ep[idx] + args.data, -1e30))
        else:
            pubt = datt = np.full(ns, -1e30)
        # step 0: blocking fetch at the start of the task
        now = t0 + 0.3                                 # dequeue, first tile
        gate = pubt[0] if not args.spec else min(pubt[0], datt[0])
        # poll until published
        if gate > now + args.travel:
            k = np.ceil((gate - now - args.travel) / args.rt)
            now = now + k * args.rt
        now += args.rt                                  # poll return / gathers
        E[0] = now + args.rt + args.tail + (step_cost(W[0]) - args.step)
        first[i] = E[0]
        prev_end = t0                                   # end of step t-2 relative to the check of step t (see below)
        for t in range(1, ns):
            ntot += 1
            c = step_cost(W[t])
            # check for step t was made at the top of step t-1 with the poll issued at the end of step t-2
            poll_sample = (E[t - 2] if t >= 2 else t0 + 0.3) + args.travel
            if pubt[t] <= poll_sample:
                E[t] = E[t - 1] + c
                continue
            if args.spec:
                # speculative gathers issued at the top of step t-1 (+fetch), sampled TRAVEL later
                spec_sample = (E[t - 2] if t >= 2 else t0 + 0.3) + args.fetch + args.travel
                if datt[t] <= spec_sample:
                    E[t] = E[t - 1] + c + 0.08          # validation of the operands
                    continue
                gate = datt[t]
            else:
                gate = pubt[t]
            nblocked += 1
            now = E[t - 1] + 0.05
            if gate > now + args.travel:
                k = np.ceil((gate - now - args.travel) / (args.rt + 0.03))
                now = now + k * (args.rt + 0.03)
            ok = now + args.rt                           # successful poll (or re-gather) has returned
            if args.spec:
                e = ok + max(args.fetch, 0.0) + args.tail + (c - args.step)
            else:
                e = ok + max(args.fetch, args.rt) + args.tail + (c - args.step)
            tblocked += e - E[t - 1] - c
            E[t] = e
        end[i] = E
        heapq.heappush(free, E[-1] + 0.3)
    span_end = max(end[i][-1] for i in ids)
    return span_end, nblocked, ntot, tblocked, end, first


t_fwd, nb0, n0, tb0, endF, _ = simulate(0, 0.0)
t_bwd, nb1, n1, tb1, endB, _ = simulate(1, t_fwd + 1.0)
nlev_note = ""
print(f"forward {t_fwd:8.1f} us  (blocked steps {nb0 / max(n0, 1):.1%}, extra {tb0 / max(nb0, 1):.2f} us each)   backward {t_bwd - t_fwd:8.1f} us  "
      f"(blocked {nb1 / max(n1, 1):.1%}, extra {tb1 / max(nb1, 1):.2f} us each)   total {t_bwd:8.1f} us{nlev_note}")
