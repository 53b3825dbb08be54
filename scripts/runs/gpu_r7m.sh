#!/bin/bash
# round 5 (second session), call m: scripts/refresh_profiles.sh on the final tree of the session
bash scripts/refresh_profiles.sh r7m
