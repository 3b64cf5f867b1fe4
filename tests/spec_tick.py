"""A second, independent restatement of the solver's specification, written from the text in plain
Python for small tables.  TEST INFRASTRUCTURE: it exists so that the C oracle (oracle/placement_oracle.c) — which *is* the
definition the GPU is held to — is itself checked against the written rules by something that shares no code with it."""
NONE = 0xFFFFFFFF
INACTIVE = 0xFFFFFFFE   # affinity of a row that is not an object: kept if placed on a live node, never placed otherwise
SAT = (1 << 64) - 1


def capacity_class(f):
    """free capacity f > 0 rounded down to three significant bits, as an ordinal: the
    water-fill takes the nodes class by class, emptiest class first, node index ascending inside a class"""
    e = f.bit_length() - 1
    return 4 * e + ((f >> (e - 2)) & 3 if e >= 2 else (f << (2 - e)) & 3)


def tick(cur, load, aff, cap, alive, rounds=2, self_assign=False):
    """self_assign (RIO_GP_CFG_REF_SELF_ASSIGN): step 2 lets a pending row claim its affinity node whether or not that node is
    alive, against the node's whole capacity (the reference's first touch asks nobody: service.rs:244-252)."""
    n, m = len(cur), len(cap)
    nxt = [NONE] * n
    used = [0] * m
    up = lambda j: j < m and bool(alive[j])
    # 1. keep (sticky): placed on a live node
    pending = []
    for i in range(n):
        if up(cur[i]):
            nxt[i] = int(cur[i])
            used[cur[i]] += int(load[i])
        else:
            pending.append(i)
    # 2. claim (first touch) with the strict prefix cut, node by node, claimants in index order
    free = [max(int(cap[j]) - used[j], 0) if (alive[j] or self_assign) else 0 for j in range(m)]
    running = [0] * m
    closed = [False] * m
    rest = []
    for i in pending:
        a = int(aff[i])
        if a == INACTIVE:
            continue                  # not an object: takes no part, stays NONE
        if not (up(a) or (self_assign and a < m)):
            rest.append(i)
            continue
        if not closed[a] and running[a] + int(load[i]) <= free[a]:
            running[a] += int(load[i])
            nxt[i] = a
        else:
            closed[a] = True          # the first overflow rejects it and everyone after it on that node
            rest.append(i)
    for j in range(m):
        used[j] += running[j]
    # 3. water-fill rounds over what is still pending, in index order
    for _ in range(rounds):
        if not rest:
            break
        fr = [max(int(cap[j]) - used[j], 0) if alive[j] else 0 for j in range(m)]
        order = sorted((j for j in range(m) if fr[j] > 0), key=lambda j: (-capacity_class(fr[j]), j))
        C = [0]
        for j in order:
            C.append(min(C[-1] + fr[j], SAT))
        q = 0
        left = []
        for i in rest:
            l = int(load[i])
            placed = False
            if order and q < C[-1]:
                k = max(t for t in range(len(order)) if C[t] <= q)
                if q + l <= C[k + 1]:
                    nxt[i] = order[k]
                    used[order[k]] += l
                    placed = True
            if not placed:
                left.append(i)
            q += l
        rest = left
    return nxt, used


def _waterfill(rest, load_of, cap, alive, used, rounds, place):
    """The water-fill rounds of section 2 step 3 over `rest` (ids in order); place(id, node) records a decision."""
    m = len(cap)
    for _ in range(rounds):
        if not rest:
            break
        fr = [max(int(cap[j]) - used[j], 0) if alive[j] else 0 for j in range(m)]
        order = sorted((j for j in range(m) if fr[j] > 0), key=lambda j: (-capacity_class(fr[j]), j))
        C = [0]
        for j in order:
            C.append(min(C[-1] + fr[j], SAT))
        q, left = 0, []
        for i in rest:
            l = load_of(i)
            ok = False
            if order and q < C[-1]:
                k = max(t for t in range(len(order)) if C[t] <= q)
                if q + l <= C[k + 1]:
                    place(i, order[k])
                    used[order[k]] += l
                    ok = True
            if not ok:
                left.append(i)
            q += l
        rest = left
    return rest


def place_pending(assign, load, cap, alive, used, idx, requester, rounds=2, self_assign=False):
    """The batched policy as include/rio_gpu_placement.h words it (service.rs:193-298 with capacity): in place on
    `assign` and `used` (python lists); returns (out_node, out_flag)."""
    m = len(cap)
    # a requested object on a node that is not alive: clean_server(that node) — ALL of its objects are un-placed
    dead = {assign[i] for i in idx if assign[i] != NONE and assign[i] < m and not alive[assign[i]]}
    found_dead = [assign[i] in dead for i in idx]   # this request found its object on a server that is not alive
    if dead:
        for r in range(len(assign)):
            if assign[r] in dead:
                assign[r] = NONE
        for d in dead:
            used[d] = 0
    free = [max(int(cap[j]) - used[j], 0) if (alive[j] or self_assign) else 0 for j in range(m)]
    run = [0] * m
    decided, how, rest = {}, {}, []
    for k, (i, r) in enumerate(zip(idx, requester)):
        if i in decided:
            continue                           # the first request for an object decides, later ones observe
        decided[i] = k
        if assign[i] != NONE:
            how[k] = "sticky"
            continue
        if alive[r] or self_assign:            # (self_assign: the requester takes its first touch whatever membership says)
            run[r] += int(load[i])             # strict prefix: an overflow stays an overflow for everyone after it
            if run[r] <= free[r]:
                how[k] = ("placed", r)
                continue
        rest.append(k)
    for k, h in how.items():
        if h != "sticky":
            used[h[1]] += int(load[idx[k]])
    spilled = {}
    _waterfill(rest, lambda k: int(load[idx[k]]), cap, alive, used, rounds, lambda k, node: spilled.__setitem__(k, node))
    for k, h in how.items():
        if h != "sticky":
            assign[idx[k]] = h[1]
    for k, node in spilled.items():
        assign[idx[k]] = node
    out_node, out_flag = [], []
    for k, (i, r) in enumerate(zip(idx, requester)):
        nd = assign[i]
        if decided[i] == k and how.get(k, None) not in (None, "sticky"):
            fl = 2                             # PLACED: first touch on the requester
        elif decided[i] == k and k in spilled:
            fl = 3                             # SPILLED
        elif nd == NONE:
            fl = 4                             # UNPLACED
        else:
            fl = 0 if nd == r else 1           # LOCAL / REDIRECT
        if decided[i] == k and found_dead[k]:
            fl |= 0x10                         # REPLACED: the server was cleaned, the object re-placed by this request
        out_node.append(nd)
        out_flag.append(fl)
    return out_node, out_flag
