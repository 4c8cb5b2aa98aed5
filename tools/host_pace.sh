# Is the HOST what paces a long run at a given object count?  Per batch: host microseconds in roft_frames_submit + roft_step against
# the batch period of the run (frames per batch x objects / value).  usage: bash tools/host_pace.sh <objects> [steps]
O=${1:-8}; S=${2:-120}
python bench.py --steps $S --warmup 3 --objects $O --windows 3 --no-cpu-baseline --pcie-frames 0 --no-extras --no-kernel-timing --json-out /tmp/host_pace.json > /dev/null 2>/tmp/host_pace.err || tail -5 /tmp/host_pace.err
python - <<PY
import json, statistics as st
d = json.load(open("/tmp/host_pace.json"))
for w in d["windows"]:
    b = w["batches"][4:]
    frames = sum(x["frames"] for x in b)
    span = (b[-1]["done_at_ms"] - b[0]["done_at_ms"]) * 1e3 / max(1, len(b) - 1)
    print("objects $O: value %8.0f | batches of %.1f frames: submit %5.1f us + step %5.1f us on the host, a batch done every %6.1f us" % (
        w["value"], frames / len(b), st.mean(x["submit_us"] for x in b), st.mean(x["step_us"] for x in b), span))
PY
