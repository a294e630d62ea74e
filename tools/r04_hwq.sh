for q in 4 8 16; do
GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline > gpurun_out/hwq_$q.log 2> gpurun_out/hwq_$q.err
python - gpurun_out/hwq_$q.log $q <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
oc=d["other_configs"]
print("HWQ", sys.argv[2], "c2", d["value"], "stream_host", oc["stream_host"].get("value"), oc["stream_host"].get("pcie",{}).get("fraction_of_pcie_ceiling"), oc["stream_host"].get("tick_latency_ms"))
print("   corpus", {k:(v["fraction_of_pcie_ceiling"], v["wall_s"]) for k,v in oc["corpus"]["legs"].items()})
PY
done
