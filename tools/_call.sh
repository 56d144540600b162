bash tools/refresh_profiles.sh r04_c 2>&1 | cut -c1-160 | tail -70
