import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end from kernels order by start").fetchall()
gaps = []
for a, b in zip(rows[:-1], rows[1:]):
    if "extract_v_kernel" in a[0] and "diag_blocks_copy" in b[0]:
        gaps.append((b[1] - a[2]) / 1e3)
    if "extract_v_kernel" in a[0] and "copyBuffer" in b[0]:
        pass
import statistics
print("extract_v -> next kernel names:", {b[0][:30] for a, b in zip(rows[:-1], rows[1:]) if "extract_v_kernel" in a[0]})
# gap from extract_v end to the first diag_blocks_copy start after it
out = []
for i, r in enumerate(rows):
    if "extract_v_kernel" in r[0]:
        for q in rows[i + 1:i + 8]:
            if "diag_blocks_copy" in q[0]:
                out.append((q[1] - r[2]) / 1e3)
                break
print("factorisation end -> gradient start: n=%d median %.1f us min %.1f us" % (len(out), statistics.median(out), min(out)))
