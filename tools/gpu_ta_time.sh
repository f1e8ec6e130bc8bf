#!/bin/bash
# timing only (no tests): TAEnv.step per mapping and size
sed -n '/^timeout -k 10 300 python - <</,$p' tools/gpu_ta.sh > /tmp/ta_time.sh
bash /tmp/ta_time.sh
