# rocgdb -batch -x tools/gpucore_report.py <python executable> <gpucore file>
# What a GPU core dump of a queue abort says: agents, queues, dispatches, the waves by kernel, and for the waves that stopped
# on a memory violation the instruction, its address operands and the kernel's arguments.
import collections
import re

import gdb

gdb.execute("set pagination off")
gdb.execute("set confirm off")
for cmd in ("info agents", "info queues", "info dispatches"):
    try:
        print("====", cmd)
        gdb.execute(cmd)
    except gdb.error as e:
        print(cmd, "->", e)
text = gdb.execute("info threads", to_string=True)
lines = text.splitlines()
print("==== info threads:", len(lines), "lines; the first 60 and every line that names a signal or a violation")
print("\n".join(lines[:60]))
odd = [ln for ln in lines if re.search(r"viol|SIG|fault|excep|abort", ln, re.I)]
print("---- lines with a signal / violation:", len(odd))
print("\n".join(odd[:80]))
by_kernel = collections.Counter()
waves = []
for th in gdb.selected_inferior().threads():
    name = th.name or ""
    if "AMDGPU" not in name and "Wave" not in name:
        continue
    waves.append(th)
print("==== waves:", len(waves))
shown = 0
for th in waves[:6000]:
    try:
        th.switch()
        fr = gdb.selected_frame()
        fn = fr.name() or "?"
        by_kernel[fn] += 1
    except gdb.error as e:
        by_kernel["(error: %s)" % str(e)[:60]] += 1
print("---- waves by innermost function")
for fn, n in by_kernel.most_common(30):
    print(n, fn)
# details of up to 6 waves per kernel whose stop reason is not a plain stop
for th in waves[:6000]:
    if shown >= 8:
        break
    try:
        th.switch()
        info = gdb.execute("thread", to_string=True)
        sig = ""
        try:
            sig = gdb.execute("p $_siginfo", to_string=True)
        except gdb.error:
            pass
        stop = gdb.execute("info program", to_string=True) if shown == 0 else ""
        pc_line = gdb.execute("x/6i $pc-16", to_string=True)
        if shown < 8:
            print("==== wave", info.strip())
            if stop:
                print(stop)
            if sig:
                print(sig[:400])
            print(pc_line)
            try:
                print(gdb.execute("info registers pc exec status trapsts mode", to_string=True)[:1500])
            except gdb.error as e:
                print("registers:", e)
            try:
                print(gdb.execute("bt 4", to_string=True)[:1500])
            except gdb.error as e:
                print("bt:", e)
            shown += 1
    except gdb.error as e:
        print("wave:", e)
