SPECS="hac:1:16384:9996 sup:1:8192:9996" bash tools/refresh_profiles.sh r04_d 2>&1 | cut -c1-160 | tail -40
