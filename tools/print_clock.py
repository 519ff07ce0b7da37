import json
d = json.load(open("gpurun_out/r2_q256_clock.json"))
for r in d["rounds"]:
    print(r["round"], r["variant"], round(r["kernel_us_hip_events"], 1), round(r["clock_ghz_median"], 3), round(r["loop_us_median"], 1), round(r["mfma_busy_frac"], 3))
