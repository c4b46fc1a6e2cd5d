#!/bin/bash
tag=$1
tools/exp/job_step.sh $tag
tools/pmc.sh $tag 2>&1 | tail -30
