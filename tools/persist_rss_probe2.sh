for pol in thread one caller; do
  echo "== RARC_IO_STREAMS=$pol"; RARC_IO_STREAMS=$pol python tools/persist_rss_probe.py 2>&1 | grep -v amdgpu.ids
done
