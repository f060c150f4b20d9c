#!/bin/bash
# rocprofv3 wrapper named in SURVEY.md §5.1:  scripts/prof_kernel.sh stats|pmc ...
#   stats [tag] [bench.py args]             -> kernel trace + per-kernel statistics (scripts/gpu_profile.sh)
#   pmc TAG H C K R stride pad fprop|dgrad|wgrad -> two PMC passes on one conv configuration (scripts/pmc_conv.sh),
#                                              then `python scripts/pmc_report.py gpurun_out/pmc/TAG`
# Counters and traces are never combined in one run; each pass stays inside the per-block slot limits.
set -o pipefail
mode=${1:-stats}; shift
here=$(dirname "$0")
case "$mode" in
  stats) exec bash "$here/gpu_profile.sh" "$@" ;;
  pmc)   exec bash "$here/pmc_conv.sh" "$@" ;;
  *) echo "usage: $0 stats|pmc ..." >&2; exit 2 ;;
esac
