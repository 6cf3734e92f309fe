#!/bin/bash
bash tools/e2e_reads_gz.sh 10 r02_e2e_gz
bash tools/r02_profiles.sh r02_prof
