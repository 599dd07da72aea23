#!/bin/bash
# Round 5 counters, part 1: RDF / S(q) / ISF entries (PMC passes) on the round's sources.
mkdir -p gpurun_out/counters
MDX_ROUND=r05 timeout -k 10 1100 python scripts/make_counters.py rdf_c2 rdf_wide rdf_c5 rdf_c1 rdf_req sq_c3 sq_default isf > gpurun_out/counters/make_counters_1.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/counters/make_counters_1.log | cut -c1-300
ls gpurun_out/counters | wc -l
