#!/bin/bash
# round-end evidence: all GPU tests, then tools/profile_r03.sh (bench lines of C2/C4/C5/C3, kernel statistics, PMC passes)
bash "$(dirname "$0")/profile_r03.sh" tests
