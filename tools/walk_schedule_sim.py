"""Replays recorded BVH walks (tools/cpu_emu/walk_trace.cpp) under different wave schedules and counts the wave-level
instructions each needs.  A walk is a fixed sequence of steps per ray (node visit / leaf of k triangles): WHICH lanes step in an
iteration is the schedule's choice, the steps themselves are not.

  classic     k_trace_shade: 64 rays of a tile start together; "while-while": node iterations until no lane holds a node, then the
              leaves' triangles (as many iterations as the largest leaf), until every lane is done
  pool        the same loop, but a lane whose walk is over takes the next ray of a pool of NB tiles at the next round
  pool-thr    pool, and the node phase ends as soon as fewer than `thr` of the lanes still hold a node (they keep it for the next round)
  greedy      every iteration runs the step kind (node / triangle / refill) that most lanes wait for

usage: walk_schedule_sim.py steps.bin [tiles per pool]
"""
import struct, sys
import numpy as np

C_NODE, C_TRI, C_LEAF, C_REFILL, C_ROUND = 130, 85, 15, 60, 12  # wave instructions per iteration of each kind (from the ISA)


def load(path):
    data = open(path, "rb").read()
    p, prim, shad = 0, [], []
    while p < len(data):
        a, b = struct.unpack_from("<II", data, p); p += 8
        prim.append(data[p:p + a]); p += a
        shad.append(data[p:p + b]); p += b
    return prim, shad


class Lane:
    __slots__ = ("seq", "p", "tri")
    def __init__(self): self.seq = b""; self.p = 0; self.tri = 0
    def start(self, seq): self.seq = seq; self.p = 0; self.tri = 0
    def done(self): return self.p >= len(self.seq) and self.tri == 0
    def wants_node(self): return self.tri == 0 and self.p < len(self.seq) and self.seq[self.p] == 0
    def wants_tri(self): return self.tri > 0 or (self.p < len(self.seq) and self.seq[self.p] != 0)
    def node(self): self.p += 1
    def tri_step(self):
        if self.tri == 0:
            self.tri = self.seq[self.p]; self.p += 1
        self.tri -= 1


def run(jobs, policy, thr=0.0, refill=True):
    """jobs: list of step sequences handed out in order to 64 lanes.  Returns (wave instructions, lane-steps, node iters, node lane-steps)."""
    lanes = [Lane() for _ in range(64)]
    nxt = 0
    cost = lane_work = n_it = n_act = t_it = t_act = 0
    def refill_lanes():
        nonlocal nxt, cost
        any_ = False
        for L in lanes:
            if L.done() and nxt < len(jobs):
                L.start(jobs[nxt]); nxt += 1; any_ = True
        if any_: cost += C_REFILL
    if not refill:  # classic: everything starts at once (jobs must be <= 64)
        for L, j in zip(lanes, jobs): L.start(j)
        nxt = len(jobs)
    while True:
        if refill: refill_lanes()
        if all(L.done() for L in lanes): break
        cost += C_ROUND
        if policy == "greedy":
            nn = sum(L.wants_node() for L in lanes); nt = sum(L.wants_tri() for L in lanes)
            if nn >= nt:
                for L in lanes:
                    if L.wants_node(): L.node()
                cost += C_NODE; n_it += 1; n_act += nn
            else:
                for L in lanes:
                    if L.wants_tri(): L.tri_step()
                cost += C_TRI; t_it += 1; t_act += nt
            continue
        # node phase
        while True:
            nn = sum(L.wants_node() for L in lanes)
            if nn == 0: break
            busy = sum(not L.done() for L in lanes)
            if thr > 0 and nn < thr * 64 and (busy - nn) > 0: break
            for L in lanes:
                if L.wants_node(): L.node()
            cost += C_NODE; n_it += 1; n_act += nn
        # leaf phase: every lane that holds a leaf works it off
        holders = [L for L in lanes if L.wants_tri()]
        if holders:
            cost += C_LEAF
            for L in holders: L.tri_step()
            cost += C_TRI; t_it += 1; t_act += len(holders)
            while True:
                rem = [L for L in holders if L.tri > 0]
                if not rem: break
                for L in rem: L.tri_step()
                cost += C_TRI; t_it += 1; t_act += len(rem)
    return cost, n_it, n_act, t_it, t_act


def report(name, tot):
    cost, n_it, n_act, t_it, t_act = tot
    print("%-22s wave instructions %9d   node iterations %7d at %4.1f %% lanes   triangle iterations %7d at %4.1f %% lanes" %
          (name, cost, n_it, 100.0 * n_act / max(1, 64 * n_it), t_it, 100.0 * t_act / max(1, 64 * t_it)))
    return cost


if __name__ == "__main__":
    prim, shad = load(sys.argv[1])
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    ntiles = len(prim) // 64
    for kind, seqs in (("camera rays", prim), ("shadow rays", shad)):
        print("==", kind, "(%d tiles, %d walks, %.1f node visits and %.2f triangles per walk)" % (
            ntiles, sum(len(s) > 0 for s in seqs), np.mean([s.count(0) for s in seqs if len(s)]), np.mean([sum(s) for s in seqs if len(s)])))
        tot = np.zeros(5, dtype=np.int64)
        for t in range(ntiles):
            tot += np.array(run(seqs[t * 64:(t + 1) * 64], "ww", refill=False))
        base = report("classic", tot)
        for policy, thr in (("ww", 0.0), ("ww", 0.25), ("ww", 0.5), ("ww", 0.75), ("greedy", 0.0)):
            tot = np.zeros(5, dtype=np.int64)
            for g in range(0, ntiles, nb):
                jobs = [s for s in seqs[g * 64:(g + nb) * 64] if len(s)]  # (the shadow job list holds only rays that need one)
                tot += np.array(run(jobs, policy, thr))
            c = report("pool %d %s thr %.2f" % (nb, policy, thr), tot)
            print("%-22s   -> %.2fx of classic" % ("", base / c))
