"""Text summary of a rocprofv3 kernel_stats.csv next to the bench line of the same command.
usage: profile_summary.py <kernel_stats.csv> <bench.json> <steps_total>"""
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
bench = json.loads(open(sys.argv[2]).read())
steps = int(sys.argv[3])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps %d --warmup %d" % (bench["steps"], bench["warmup"]))
print("#   N=1, c3 workload (B=512), %d warm-up + %d timed steps = %d steps; bench line of the un-profiled run of the same command: %.2f ms/step"
      % (bench["warmup"], bench["steps"], steps, bench["ms_per_step"]))
print("# total kernel time %.1f ms; per-step figures divide by %d (the 'secondary' GEMM / attention timings and model" % (tot / 1e6, steps))
print("# construction add a few launches outside the steps).  The backward runs on three HIP streams (weight-gradient")
print("# GEMMs and the SOM backward next to the main chain): overlapping kernels share the GPU, so their individual")
print("# durations stretch and the per-step column sums to more than the wall-clock step.\n")
for r in rows[:36]:
    print(f"{r['Name'][:88]:88s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:8.1f} us  per-step {float(r['TotalDurationNs'])/1e6/steps:7.3f} ms  {float(r['Percentage']):5.1f}%")
