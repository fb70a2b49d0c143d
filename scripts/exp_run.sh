for a in "--evict 600 --retouch" "--evict 600 --retouch --prewarm-code"; do echo "== kbench $a"; python scripts/kbench.py $a 2>&1 | grep -v amdgpu | cut -c1-110; done
