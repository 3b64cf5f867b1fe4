"""A second, independent restatement of the solver's specification, written from the text in plain
Python for small tables.  TEST INFRASTRUCTURE: it exists so that the C oracle (oracle/placement_oracle.c) — which *is* the
definition the GPU is held to — is itself checked against the written rules by something that shares no code with it."""
NONE = 0xFFFFFFFF
SAT = (1 << 64) - 1


def tick(cur, load, aff, cap, alive, rounds=2):
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
    free = [max(int(cap[j]) - used[j], 0) if alive[j] else 0 for j in range(m)]
    running = [0] * m
    closed = [False] * m
    rest = []
    for i in pending:
        a = int(aff[i])
        if not up(a):
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
        order = sorted((j for j in range(m) if fr[j] > 0), key=lambda j: (-fr[j], j))
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
