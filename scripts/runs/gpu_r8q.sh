#!/bin/bash
# round 6, call q: refresh of everything profiles/ holds, on the current tree
bash scripts/refresh_profiles.sh r8q_refresh
