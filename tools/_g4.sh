cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
(time timeout 2400 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -15) 2>&1
python3 -c "import __graft_entry__ as g; g.smoke()"
