import json,sys
for f in sys.argv[1:]:
    try:
        b=json.loads(open(f).read().strip().splitlines()[-1])
        k=b['kernels']
        print(f, b['ms_per_step'], b['dtype'], 'launches', sum(v['launches_per_step'] for v in k.values()), 'loss', b['config'].get('loss'))
    except Exception as e:
        print(f, 'ERR', e)
