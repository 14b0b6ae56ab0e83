#!/bin/bash
bash "$GRAFT_REPO_ROOT/tools/profile_r03.sh" "$@"
