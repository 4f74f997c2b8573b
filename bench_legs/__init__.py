"""bench.py's legs: one module per workload, each `run(ctx)` fills ctx.result / ctx.others (see bench.py)."""
