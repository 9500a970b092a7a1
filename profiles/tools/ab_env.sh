#!/bin/bash
# Same-box A/B of one build under two environments: alternate A (plain), B (with the given VAR=VALUE), A, B ...
#   profiles/tools/ab_env.sh WSAE_STRIP_PREDICT=0 [rounds] [extra bench.py args]
set -e
KV=$1; ROUNDS=${2:-2}; shift; shift || true
mkdir -p gpurun_out
python3 - "$KV" "$ROUNDS" "$@" <<'PY'
import json, os, subprocess, sys
kv, rounds, extra = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
k, v = kv.split("=", 1)
for r in range(rounds):
    for tag, env in (("A", dict(os.environ)), ("B", dict(os.environ, **{k: v}))):
        base = [sys.executable, "bench.py", "--no-cpu-baseline", "--steps", "100", "--warmup", "100"] + extra
        o = subprocess.run(base + ["--profile-all"], env=env, capture_output=True, text=True)
        line = [l for l in o.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(tag, "FAILED", o.stderr[-800:]); continue
        j = json.loads(line[-1])
        ks = {n.split("<")[0].replace("_kernel", ""): round(t * 1e3, 1) for n, t in j["kernel_ms_per_step"].items()}
        o2 = subprocess.run(base, env=env, capture_output=True, text=True)
        j2 = json.loads([l for l in o2.stdout.splitlines() if l.startswith("{")][-1])
        print(tag, kv if tag == "B" else "", "step_us", round(j2["ms_per_step"] * 1e3, 1), "loss", j2["final_loss"], ks, flush=True)
PY
