"""Static model of workgroup dispatch on MI355X for one-workgroup-per-(sample, head) launches on padded batches (see skf_common.h: skf_deal_rank;\nprofiles/r05o_attn_order.txt): makespan of the (sample, head) numbering with / without XCD-contiguous ids, of the sorted deal, and two bounds."""
import numpy as np, heapq
rng = np.random.default_rng(0)
def sim(costs_by_bid, slots_per_se=16):
    # bid -> xcd = bid%8, arrival i = bid//8, se = i%4; in-order list scheduling per SE on `slots_per_se` slots
    n = len(costs_by_bid); fin = 0.0
    for x in range(8):
        for se in range(4):
            q = [costs_by_bid[b] for b in range(n) if b % 8 == x and (b // 8) % 4 == se]
            h = [0.0] * slots_per_se
            for c in q:
                t = heapq.heappop(h); heapq.heappush(h, t + c)
            fin = max(fin, max(h))
    return fin
def ideal(costs, slots=512): return max(costs.sum() / slots, costs.max())
B, H, L = 128, 8, 200
res = {k: [] for k in ("remap", "plain", "sorted-deal", "ideal", "dyn-global")}
for trial in range(20):
    n = np.clip(np.rint(rng.normal(80, 35, B)), 2, L).astype(int)
    t = np.ceil(n / 16)
    cost = 7.6 + 24.0 * t * t / 169.0          # per (b,h) workgroup, us
    # remap: lid = xcd*128 + i
    def c_remap(bid): lid = (bid % 8) * 128 + bid // 8; return cost[lid // 8]
    def c_plain(bid): return cost[bid // 8]
    order = np.argsort(-cost, kind="stable")
    def c_sorted(bid):
        x, i = bid % 8, bid // 8; k = 32 * (i // 4) + 4 * x + i % 4; return cost[order[k // 8]]
    for name, f in (("remap", c_remap), ("plain", c_plain), ("sorted-deal", c_sorted)):
        res[name].append(sim([f(b) for b in range(B * H)]))
    allc = np.repeat(cost, 8)
    res["ideal"].append(ideal(allc))
    h = [0.0] * 512
    for c in allc: tt = heapq.heappop(h); heapq.heappush(h, tt + c)
    res["dyn-global"].append(max(h))
for k, v in res.items(): print("%-12s %.1f us" % (k, np.mean(v)))
