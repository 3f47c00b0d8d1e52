#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python tools/adj_rows_check.py time 2>&1 | tail -40
