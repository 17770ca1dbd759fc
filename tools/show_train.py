import json, sys
r = json.load(open(sys.argv[1]))
print(r["value"], r["ms_per_step"], r["config"]["graph_state"], r["config"]["final_losses"])
tot = 0
for k, v in list(r["roofline"]["per_kernel"].items())[:int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    print("%8.3f ms %4d  %8.1f TF  %8.1f GB/s  %s" % (v["ms"], v["launches"], v["tflops"], v["GBs"], k))
print("profiled total", sum(v["ms"] for v in r["roofline"]["per_kernel"].values()))
