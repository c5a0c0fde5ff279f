# save / load rates of a 10M x 768 fp16 shard (15.4 GB): O_DIRECT vs buffered, page cache hot vs dropped, 8 / 16 threads
mkdir -p gpurun_out
for args in "" "--cold" "--direct-load" "--direct-load --cold" "--no-direct" "--threads 16" "--threads 4" "--threads 2"; do
  echo "== $args"
  python tools/persist_probe.py --rows 10000000 --dim 768 --dir /tmp/rarc_rates --verify 0 $args 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('save %.2f GB/s (wall %.2fs, direct %d)  load %.2f GB/s (wall %.2fs)  identical %s  hwm_end %d MB  rss_start %d MB' % (o['save']['gb_per_s'], o['save_wall_s'], o['save']['direct'], o['load']['gb_per_s'], o['load_wall_s'], o['search_identical'], o['hwm_kb_end']//1024, o['rss_kb_start']//1024))"
done 2>&1 | tee gpurun_out/r04_persist_rates.txt
echo "== /dev/shm (tmpfs: O_DIRECT refused -> buffered)"; python tools/persist_probe.py --rows 10000000 --dim 768 --dir /dev/shm/rarc_rates --verify 0 2>/dev/null | tail -1 | cut -c1-900 | tee -a gpurun_out/r04_persist_rates.txt
