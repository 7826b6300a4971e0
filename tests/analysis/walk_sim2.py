#!/usr/bin/env python3
"""Offline model (analysis only, CPU oracle traces via walk_sim.py) of TWO WALK JOBS PER LANE with kind-sorted phases — round 5's experiment
(generalised: the two slots of a lane hold independent paths, each running shadow -> closest as today).

Per trip of a wave: a BOX phase (every lane advances one of its slots that stands at a box) and a TRIANGLE phase that runs only when at least `tri_min`
lanes have a slot at a triangle (or no lane is at a box).  Costs = instructions of the current kernel's blocks (tools/asm_loop.py: mixed trip ~217 = box 77 +
triangle 97 + fetch / loop 43), plus the selects a two-slot lane pays to pick the slot a phase works on.

  python tests/analysis/walk_sim2.py [scene] [strips]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import walk_sim as ws

C_BOX, C_TRI, C_FETCH, C_LOOP = 77, 97, 33, 10
SEL_BOX, SEL_TRI = 20, 22
C_BATCH, C_SETUP = 120, 200


def sim_one(seqs, batch=24, inner=8):
    """today's kernel: one walk job per lane, both bodies per trip when both kinds are present"""
    cost = visits = 0
    nxt = 0
    n = len(seqs)
    cur = [None] * 64
    pos = [0] * 64
    st = [0] * 64            # 0 empty, 1 walking, 2 switch, 3 done
    lane_instr = 0
    while True:
        walking = sum(1 for s in st if s == 1)
        work = any(s in (2, 3) for s in st)
        if walking == 0 or (64 - walking >= batch and (work or nxt < n)):
            cost += C_BATCH
            setup = False
            for l in range(64):
                if st[l] == 3: st[l] = 0
                if st[l] == 0 and nxt < n:
                    cur[l] = seqs[nxt]; nxt += 1; pos[l] = 0; st[l] = 1; setup = True
                elif st[l] == 2:
                    pos[l] += 1; st[l] = 1; setup = True
                while st[l] == 1:
                    if pos[l] >= len(cur[l]): st[l] = 3
                    elif cur[l][pos[l]] == 3: st[l] = 2
                    elif cur[l][pos[l]] == 0: visits += 1; pos[l] += 1; continue      # a terminator: fetched, not tested
                    break
            if setup: cost += C_SETUP
            if not any(s == 1 for s in st):
                if nxt >= n and not any(s in (2, 3) for s in st): break
                continue
        for _ in range(inner):
            nb = sum(1 for l in range(64) if st[l] == 1 and cur[l][pos[l]] == 1)
            nt = sum(1 for l in range(64) if st[l] == 1 and cur[l][pos[l]] == 2)
            if nb + nt == 0: break
            cost += C_LOOP + C_FETCH + (C_BOX if nb else 0) + (C_TRI if nt else 0)
            lane_instr += nb * C_BOX + nt * C_TRI + (nb + nt) * (C_FETCH + C_LOOP)
            for l in range(64):
                if st[l] == 1:
                    visits += 1; pos[l] += 1
                    while st[l] == 1:
                        if pos[l] >= len(cur[l]): st[l] = 3
                        elif cur[l][pos[l]] == 3: st[l] = 2
                        elif cur[l][pos[l]] == 0: visits += 1; pos[l] += 1; continue
                        break
    return cost, visits, lane_instr / max(1, cost) / 64.0


def sim_two(seqs, W=2, batch=24, inner=8, tri_min=32, sel=True):
    cost = visits = 0
    nxt = 0
    n = len(seqs)
    L = 64 * W
    cur = [None] * L
    pos = [0] * L
    st = [0] * L
    last = [0] * 64
    trips = [0, 0]
    lanes = [0, 0]

    def settle(v):
        nonlocal visits
        while st[v] == 1:
            if pos[v] >= len(cur[v]): st[v] = 3
            elif cur[v][pos[v]] == 3: st[v] = 2
            elif cur[v][pos[v]] == 0: visits += 1; pos[v] += 1; continue
            break

    while True:
        walking = sum(1 for s in st if s == 1)
        work = any(s in (2, 3) for s in st)
        if walking == 0 or (L - walking >= batch * W and (work or nxt < n)):
            cost += C_BATCH * W
            setup = False
            for v in range(L):
                if st[v] == 3: st[v] = 0
                if st[v] == 0 and nxt < n:
                    cur[v] = seqs[nxt]; nxt += 1; pos[v] = 0; st[v] = 1; setup = True
                elif st[v] == 2:
                    pos[v] += 1; st[v] = 1; setup = True
                settle(v)
            if setup: cost += C_SETUP * W
            if not any(s == 1 for s in st):
                if nxt >= n and not any(s in (2, 3) for s in st): break
                continue
        for _ in range(inner):
            moved = False
            for kind, cbody, csel in ((1, C_BOX, SEL_BOX), (2, C_TRI, SEL_TRI)):
                picks = []
                for l in range(64):
                    c = [v for v in range(l * W, l * W + W) if st[v] == 1 and cur[v][pos[v]] == kind]
                    if c:
                        # prefer the slot that was not advanced last (its fetch has had time)
                        c.sort(key=lambda v: (v % W) == last[l])
                        picks.append(c[0])
                if not picks: continue
                if kind == 2 and len(picks) < tri_min:
                    nb = sum(1 for l in range(64) if any(st[v] == 1 and cur[v][pos[v]] == 1 for v in range(l * W, l * W + W)))
                    if nb > 0: continue          # wait for company while box work remains
                cost += cbody + (csel if sel else 0) + C_FETCH
                trips[kind - 1] += 1; lanes[kind - 1] += len(picks)
                moved = True
                for v in picks:
                    visits += 1; pos[v] += 1; last[v // W] = v % W
                    settle(v)
            cost += C_LOOP
            if not moved: break
    return cost, visits, trips, lanes


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "dragon"
    strips = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    trace, p = ws.record(name, strips)
    bounces = ws.parse(trace, p)
    limit = int(os.environ.get("SIM_PATHS", "6000"))
    for b, items in enumerate(bounces):
        mid = max(0, len(items) // 2 - limit // 2)
        items = items[mid:mid + limit]
        seqs = [s.tolist() for s in ws.lane_sequences(items)]
        ops = np.concatenate([np.asarray(s, np.uint8) for s in seqs])
        nb, nt = int((ops == 1).sum()), int((ops == 2).sum())
        print("bounce %d: %d paths, boxes %.2f triangles %.2f of %d tests" % (b, len(seqs), nb / (nb + nt), nt / (nb + nt), nb + nt))
        c0, v0, u0 = sim_one(seqs)
        print("   one job per lane (today)                  : %6.1f instr / visit   (lane utilisation of the trips %.2f)" % (c0 / v0, u0))
        for W, tm in ((2, 1), (2, 16), (2, 24), (2, 32), (2, 40), (3, 32), (4, 32), (4, 48)):
            c, v, trips, lanes = sim_two(seqs, W=W, tri_min=tm)
            print("   %d jobs per lane, triangle phase at >= %2d   : %6.1f instr / visit   (box trips %d x %.0f lanes, tri trips %d x %.0f lanes)   %+.0f %%" % (
                W, tm, c / v, trips[0], lanes[0] / max(1, trips[0]), trips[1], lanes[1] / max(1, trips[1]), 100.0 * (c / v / (c0 / v0) - 1)))


if __name__ == "__main__":
    main()
