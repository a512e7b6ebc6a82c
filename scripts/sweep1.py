run('default')
for skip in (1, 2, 4, 3, 5, 6, 7):
    run(f'skip={skip}', MI355_DBG_SKIP=skip)
for nt in (256, 512, 1024):
    for vpt in (4, 16):
        run(f'nt={nt} vpt={vpt}', MI355_SCAN_THREADS=nt, MI355_SCAN_VPT=vpt)
        run(f'nt={nt} vpt={vpt} notopk', MI355_SCAN_THREADS=nt, MI355_SCAN_VPT=vpt, MI355_DBG_SKIP=4)
