#!/bin/bash
# tools/gpu_pmc_all.sh -- PMC passes + kernel traces of every workload profiles/pmc_traffic.json covers (PMC_TAG names the output set)
bash tools/gpu_pmc_report.sh ${PMC_TAG:-r04_v1} mulrelin_n8192
bash tools/gpu_pmc_report.sh ${PMC_TAG:-r04_v1} mulrelin_n16384 --n 16384 --batch 1024
bash tools/gpu_pmc_report.sh ${PMC_TAG:-r04_v1} ntt_n8192 --workload ntt
bash tools/gpu_pmc_report.sh ${PMC_TAG:-r04_v1} mulrelin_n8192_bits54-54-54-56 --coeff-bits 54,54,54,56
bash tools/gpu_pmc_report.sh ${PMC_TAG:-r04_v1} pir_n16384 --workload pir --n 16384 --batch 256 --pir-rows 512
