export TMPDIR=/tmp
O=gpurun_out/${TAG:-r05t}; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x -k "physics_only_b4096 or host_episode or time_limit_and_auto" 2>&1 | tail -5 > $O/pytest.txt
