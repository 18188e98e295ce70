python tools/step_ab.py --T 13 --rounds 9 --steps 5 --ab "wide_persist_min_x10=20,wide_persist_min_x10=10,wide_persist_min_x10=15" 2>&1 | grep -v amdgpu
python tools/step_ab.py --T 25 --rounds 5 --steps 5 --ab "wide_persist_min_x10=20,wide_persist_min_x10=10" 2>&1 | grep -v amdgpu
python tools/step_ab.py --T 50 --rounds 5 --steps 3 --ab "wide_persist_min_x10=20,wide_persist_min_x10=10" 2>&1 | grep -v amdgpu
