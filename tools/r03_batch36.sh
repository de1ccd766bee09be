#!/bin/bash
RUN_NAME=r03_v3 bash tools/run_profile_set.sh
