run('default')
run('notopk', MI355_DBG_SKIP=4)
for nt in (512, 1024):
    for vpt in (4, 16):
        run(f'nt={nt} vpt={vpt}', MI355_SCAN_THREADS=nt, MI355_SCAN_VPT=vpt)
