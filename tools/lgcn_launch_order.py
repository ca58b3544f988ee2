"""Per-launch HIP-event times of a LightGCN step at the Yelp2018 shape IN LAUNCH ORDER (the two dense layers of a step carry the
same name), and the same step on an all-zero ego table: a dense layer over dE (zero outside the batch) takes 3-4 us less than the
forward layer over T on either table -- position in the step, not the data (r05: 45.7 / 42.1 us with the events' own ~3.5 us)."""
import os, sys, json
sys.argv = [sys.argv[0], "yelp2018"]
src = open("tools/bench_lgcn.py").read().split("ops.timing_begin()")[0]
exec(src)
def order(tag):
    ops.timing_begin()
    for k in range(10):
        state.step(ops.LOSS_RUBIBCEBOTH, batches[k, 0], batches[k, 1], batches[k, 2])
    seq = ops.timing_end(512)
    per = len(seq) // 10
    names = [n for n, _ in seq[:per]]
    avg = [round(1e3 * sum(seq[s * per + i][1] for s in range(10)) / 10, 1) for i in range(per)]
    print(tag, list(zip(names, avg)), flush=True)
order("trained tables")
# the same step on an all-zero ego table (data dependence of a dense layer's time: zero rows gathered)
state.T.zero_()
order("zero ego table")
