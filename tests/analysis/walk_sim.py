#!/usr/bin/env python3
"""Offline model of walk-kernel scheduling policies (analysis only, CPU oracle).

Records, with the oracle's trace hook, the entry types every bounce walk of a sample of the frame visits, then replays
them through models of one wave (64 lanes) under different stepping policies and prints the modelled issue slots per
entry visited.  Costs are VALU issue slots of the current kernel's blocks (build/asm): box test, triangle test, entry
fetch, per-trip scheduler overhead, batch (fold + refill + setup).

  python tests/analysis/walk_sim.py [scene] [strips]"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "web-ray-tracer_amd"))
import flx_oracle
from flexlight_hip.scene_io import Scene

C_BOX, C_TRI, C_FETCH, C_SCHED, C_BATCH, C_SETUP = 64, 85, 25, 20, 120, 200


def record(name, strips):
    sc = Scene.golden(name)
    lib = flx_oracle.lib()
    lib.flx_oracle_set_trace.argtypes = [C.c_void_p, C.c_size_t]
    lib.flx_oracle_trace_length.restype = C.c_size_t
    p = sc.frame_params(use_filter=0)
    total_strips = (p.height + 7) // 8
    p.tile_rows, p.tile_count, p.tile_index = 8, max(1, total_strips // strips), 0
    buf = np.zeros(1 << 30, np.uint8)
    lib.flx_oracle_set_trace(buf.ctypes.data, buf.size)
    t = time.time()
    flx_oracle.render(sc, p, threads=1)
    n = lib.flx_oracle_trace_length()
    lib.flx_oracle_set_trace(None, 0)
    print("trace: %d bytes in %.1f s (%d strips of 8 rows)" % (n, time.time() - t, (total_strips + p.tile_count - 1) // p.tile_count), file=sys.stderr)
    return buf[:n].copy(), p


def parse(trace, p):
    """-> per bounce: list of (item order key, shadow ops or None, closest ops) with ops = uint8 arrays of types"""
    hdr = np.flatnonzero(trace >= 0xF0)
    hdr = hdr[trace[hdr] != 0xFF]
    # headers are 8 bytes; bytes inside headers (px/py) may be >= 0xF0 too: walk the stream instead
    walks = {}
    i, n = 0, trace.size
    ends = np.flatnonzero(trace == 0xFF)
    S = p.samples
    tiles_x = (p.width + 7) // 8
    while i < n:
        kind = int(trace[i]) & 1
        b, s = int(trace[i + 1]), int(trace[i + 2])
        px = int(trace[i + 4]) | int(trace[i + 5]) << 8
        py = int(trace[i + 6]) | int(trace[i + 7]) << 8
        j = i + 8
        e = ends[np.searchsorted(ends, j)]
        ops = trace[j:e]
        row = p.height - 1 - py
        strip = row // 8
        key = (((strip * tiles_x + px // 8) * S + s) << 6) | ((row % 8) * 8 + px % 8)
        walks.setdefault(b, {}).setdefault(key, [None, None])[kind] = ops
        i = e + 1
    out = []
    for b in sorted(walks):
        items = [(k, v[0], v[1]) for k, v in sorted(walks[b].items())]
        out.append(items)
    return out


def lane_sequences(items):
    """one op sequence per path: shadow walk, a SWITCH marker (3), closest walk (entry kinds only, for sim_current / sim_slots)"""
    seqs = []
    for _, sh, cl in items:
        parts = []
        if sh is not None:
            parts += [sh & 15, np.array([3], np.uint8)]
        parts.append(cl & 15 if cl is not None else np.zeros(0, np.uint8))
        seqs.append(np.concatenate(parts))
    return seqs


K_BOX, K_TRI, K_XF, K_NEW = 0, 1, 2, 3


def op_sequences(items):
    """per path, the tests the queue schedulers run: NEW (refill + ray set-up), then per walk its box / triangle tests with an
    XFORM before every entry whose transform differs from the cached one, NEW again between the shadow and the closest walk"""
    seqs = []
    for _, sh, cl in items:
        ops = [K_NEW]
        for walk in (sh, cl):
            if walk is None:
                continue
            if len(ops) > 1:
                ops.append(K_NEW)
            for byte in walk.tolist():
                kind = byte & 15
                if kind == 0:
                    continue                      # the terminator is handled when its link is routed
                if byte & 0x10:
                    ops.append(K_XF)
                ops.append(K_BOX if kind == 1 else K_TRI)
        seqs.append(ops)
    return seqs


def sim_local_queues(seqs, walks=128, overhead=55, costs=(64, 85, 110, 300), min_tri=0):
    """one wave owning `walks` walks in LDS, four wave-local queues; each trip runs ONE kind of test on up to 64 queued walks"""
    n = len(seqs)
    nxt = 0
    queues = [[], [], [], []]
    pos = {}
    live = 0
    cost = 0
    visits = 0
    trips = [0, 0, 0, 0]
    lanes = [0, 0, 0, 0]
    # a NEW op both retires a path and starts the next one in the same slot: model the slot pool by seeding `walks` paths
    for _ in range(min(walks, n)):
        pos[nxt] = 0; queues[K_NEW].append(nxt); nxt += 1
    while any(queues):
        sizes = [len(q) for q in queues]
        k = max(range(4), key=lambda q: sizes[q])
        if min_tri and k == K_TRI and sizes[K_TRI] < min_tri and sizes[K_BOX] > 0:
            k = K_BOX
        batch, queues[k] = queues[k][:64], queues[k][64:]
        cost += costs[k] + overhead
        trips[k] += 1; lanes[k] += len(batch)
        for w in batch:
            if k in (K_BOX, K_TRI): visits += 1
            pos[w] += 1
            ops = seqs[w]
            if pos[w] >= len(ops):
                del pos[w]
                if nxt < n:                       # the fold of this path is the NEW op of the next one
                    pos[nxt] = 0; queues[K_NEW].append(nxt); nxt += 1
            else:
                queues[ops[pos[w]]].append(w)
    return cost, visits, trips, lanes


def sim_current(seqs, inner=4, batch=16, box_run=0):
    """k_wf_walk_pre: every walking lane steps one entry per trip (box and triangle code both run when both kinds are present)."""
    cost = 0
    visits = 0
    nxt = 0
    cur = [None] * 64
    pos = np.zeros(64, np.int64)
    state = np.zeros(64, np.int8)       # 0 empty, 1 walking, 2 parked (switch), 3 done
    n = len(seqs)
    while True:
        walking = state == 1
        parked = 64 - walking.sum()
        can_refill = nxt < n
        work = ((state == 2) | (state == 3)).any()
        if walking.sum() == 0 or (parked >= batch and (work or can_refill)):
            cost += C_BATCH
            setup = False
            for l in range(64):
                if state[l] == 3: state[l] = 0
                if state[l] == 0 and nxt < n:
                    cur[l] = seqs[nxt]; nxt += 1; pos[l] = 0; state[l] = 1; setup = True
                elif state[l] == 2:
                    pos[l] += 1; state[l] = 1; setup = True
                if state[l] == 1 and pos[l] >= cur[l].size: state[l] = 3
                if state[l] == 1 and cur[l][pos[l]] == 3: state[l] = 2
            if setup: cost += C_SETUP
            if (state == 1).sum() == 0:
                if nxt >= n and not ((state == 2) | (state == 3)).any(): break
                continue
        for _ in range(inner):
            types = np.zeros(64, np.int8)
            for l in range(64):
                if state[l] == 1: types[l] = cur[l][pos[l]]
            anyb, anyt = (types == 1).any(), (types == 2).any() or ((types == 0) & (state == 1)).any()
            cost += C_SCHED + (C_BOX if anyb else 0) + (C_TRI if anyt else 0) + C_FETCH
            for l in range(64):
                if state[l] == 1:
                    visits += 1
                    pos[l] += 1
                    if pos[l] >= cur[l].size: state[l] = 3
                    elif cur[l][pos[l]] == 3: state[l] = 2
    return cost, visits


def sim_slots(seqs, W=2, overhead=45, batch_frac=0.25):
    """W walks per lane kept in LDS; each trip the wave runs ONE test kind, on one ready walk per lane."""
    cost = 0
    visits = 0
    nxt = 0
    n = len(seqs)
    L = 64 * W
    cur = [None] * L
    pos = np.zeros(L, np.int64)
    state = np.zeros(L, np.int8)
    lane = np.arange(L) // W
    while True:
        idle = (state != 1).sum()
        if (state == 1).sum() == 0 or (idle >= L * batch_frac and (nxt < n or ((state == 2) | (state == 3)).any())):
            cost += C_BATCH
            setup = False
            for l in range(L):
                if state[l] == 3: state[l] = 0
                if state[l] == 0 and nxt < n:
                    cur[l] = seqs[nxt]; nxt += 1; pos[l] = 0; state[l] = 1; setup = True
                elif state[l] == 2:
                    pos[l] += 1; state[l] = 1; setup = True
                if state[l] == 1 and pos[l] >= cur[l].size: state[l] = 3
                if state[l] == 1 and cur[l][pos[l]] == 3: state[l] = 2
            if setup: cost += C_SETUP * W
            if (state == 1).sum() == 0:
                if nxt >= n and not ((state == 2) | (state == 3)).any(): break
                continue
        for _ in range(4):
            types = np.zeros(L, np.int8)
            for l in range(L):
                if state[l] == 1: types[l] = 1 if cur[l][pos[l]] == 1 else 2
            boxLanes = np.unique(lane[types == 1]).size
            triLanes = np.unique(lane[types == 2]).size
            if boxLanes == 0 and triLanes == 0: break
            # pick the kind with the better lanes-per-slot ratio
            kind = 1 if boxLanes * (C_TRI + overhead) >= triLanes * (C_BOX + overhead) else 2
            cost += (C_BOX if kind == 1 else C_TRI) + overhead
            done_lane = set()
            for l in range(L):
                if types[l] == kind and lane[l] not in done_lane:
                    done_lane.add(lane[l])
                    visits += 1
                    pos[l] += 1
                    if pos[l] >= cur[l].size: state[l] = 3
                    elif cur[l][pos[l]] == 3: state[l] = 2
    return cost, visits


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "dragon"
    strips = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    trace, p = record(name, strips)
    bounces = parse(trace, p)
    for b, items in enumerate(bounces):
        limit = int(os.environ.get("SIM_PATHS", "20000"))
        mid = max(0, len(items) // 2 - limit // 2)
        items = items[mid:mid + limit]                     # the middle of the sampled strips: dragon, not sky
        seqs = lane_sequences(items)
        ops = np.concatenate(seqs)
        nb, nt = int((ops == 1).sum()), int((ops == 2).sum() + (ops == 0).sum())
        qs = op_sequences(items)
        nx = sum(o.count(K_XF) for o in qs)
        print("bounce %d: %d paths, %d visits (box %.2f tri %.2f), %.1f transform changes per path" % (b, len(seqs), nb + nt, nb / (nb + nt), nt / (nb + nt), nx / len(qs)))
        ideal = (nb * (C_BOX + 55) + nt * (C_TRI + 55)) / 64.0
        c0, v0 = sim_current(seqs)
        print("   one walk per lane (k_wf_walk_pre)     : %5.2f slots / visit" % (c0 / v0))
        for walks in (128, 256, 1280):
            c, v, trips, lanes = sim_local_queues(qs, walks)
            fill = " ".join("%s %.0f" % (nm, lanes[k] / max(1, trips[k])) for k, nm in enumerate(("box", "tri", "xf", "new")))
            print("   queues over %4d walks                 : %5.2f slots / visit   (mean batch: %s)" % (walks, c / v, fill))
        c, v, trips, lanes = sim_local_queues(qs, 128, min_tri=40)
        print("   queues over  128 walks, tri >= 40      : %5.2f slots / visit" % (c / v))
        print("   perfect regrouping, no set-up ops      : %5.2f slots / visit" % (ideal / (nb + nt)))


if __name__ == "__main__":
    main()
