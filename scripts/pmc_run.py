run('default')
