#!/bin/bash
# round 6: rocprofv3 summaries of the bench's dominant kernels (kernel trace + separate PMC passes each)
bash tools/profile_round.sh r06 c2 f64 2>&1 | tail -12
bash tools/profile_round.sh r06 c2 f32 2>&1 | tail -8
bash tools/profile_round.sh r06sml c3 f64 --c3-small 2>&1 | tail -8
bash tools/profile_round.sh r06 c3 f64 2>&1 | tail -8
