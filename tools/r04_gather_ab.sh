# the default line's corpus routes, repeated, with the staging stream at normal / high priority
for rep in 1 2; do for prio in 0 1; do
SILERO_VAD_AMD_STAGE_PRIO=$prio python bench.py --no-cpu-baseline > gpurun_out/line_p$prio.log 2>gpurun_out/line_p$prio.err || tail -3 gpurun_out/line_p$prio.err
python - gpurun_out/line_p$prio.log $prio <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
oc=d["other_configs"]
print("stage prio", sys.argv[2], {k: v["fraction_of_pcie_ceiling"] for k, v in oc["corpus"]["legs"].items()}, "stream_host", oc["stream_host"]["pcie"]["fraction_of_pcie_ceiling"], oc["stream_host_8k"]["pcie"]["fraction_of_pcie_ceiling"])
PY
done; done
