"""Shape of the captured step's DAG (tuning aid): MOPOE_GRAPH_DOT=/tmp/step.dot python bench.py --steps 2 --warmup 1
--no-cpu-baseline --no-roofline, then python tests/tools/graph_dag.py /tmp/step.dot -> nodes, edges, longest chain,
and for every node with more than one predecessor or successor its neighbours (the forks and joins)."""
import re, sys, collections
txt = open(sys.argv[1]).read()
labels = {}
for m in re.finditer(r'"?(\w+)"?\s*\[([^\]]*)\]', txt):
    lab = re.search(r'label="([^"]*)"', m.group(2))
    if lab and "->" not in m.group(0):
        labels[m.group(1)] = lab.group(1).replace("\\n", " ")[:90]
edges = re.findall(r'"?(\w+)"?\s*->\s*"?(\w+)"?', txt)
succ, pred = collections.defaultdict(list), collections.defaultdict(list)
for a, b in edges:
    succ[a].append(b); pred[b].append(a)
nodes = sorted(set(labels) | set(succ) | set(pred))
print(len(nodes), "nodes", len(edges), "edges")
# longest chain (in nodes) by memoised DFS over the DAG
sys.setrecursionlimit(100000)
depth = {}
def d(n):
    if n not in depth:
        depth[n] = 1 + max((d(p) for p in pred[n]), default=0)
    return depth[n]
longest = max(d(n) for n in nodes)
print("longest chain:", longest, "nodes; width profile (nodes per depth level), first 60 levels with width > 1:")
lv = collections.Counter(depth.values())
print(" ".join(f"{k}:{v}" for k, v in sorted(lv.items()) if v > 1)[:3000])
print("forks / joins:")
order = sorted(nodes, key=lambda n: depth[n])
for n in order:
    if len(succ[n]) > 1 or len(pred[n]) > 1:
        print(f"  depth {depth[n]:4d} {labels.get(n, n)[:70]:70s} preds {len(pred[n])} succs {len(succ[n])}")
