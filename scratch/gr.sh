#!/bin/bash
# usage: scratch/gr.sh <timeout> '<command>'   -- retries only while no box is free (exit code 3)
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout "$1" -- "$2" > /tmp/gr_last.txt 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then tail -4 /tmp/gr_last.txt; exit $rc; fi
  sleep 40
done
echo "no box after 20 tries"; exit 3
