cd $GRAFT_REPO_ROOT
timeout 300 python tools/cert_by_tile.py 2>&1 | grep -E "tile \(|DEBUG" | cut -c1-260 | head -40
