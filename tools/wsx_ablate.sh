for f in "" "-DSKF_WSX_ABLATE_MFMA=1" "-DSKF_WSX_ABLATE_STORE=1" "-DSKF_WSX_ABLATE_LOAD=1" "-DSKF_WSX_ABLATE_STORE=1 -DSKF_WSX_ABLATE_LOAD=1" "-DSKF_WSX_ABLATE_STORE=1 -DSKF_WSX_ABLATE_LOAD=1 -DSKF_WSX_ABLATE_MFMA=1" "-DSKF_WSX_ABLATE_SPLIT=1"; do
  export SKF_EXTRA_HIPCC_FLAGS="$f"; python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  echo "== flags: $f"; timeout 300 python tools/wsx_check.py 2>&1 | grep "B\[K\]\[N\]" | awk -F'|' '{print $1 $3}' | cut -c1-50
done
