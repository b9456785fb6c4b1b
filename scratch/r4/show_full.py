import json, sys
for f in sys.argv[1:]:
    d=json.load(open(f))
    print("==",f)
    for k,r in d['runs'].items():
        t=r['train']; e=r['eval']; m=e['picks_vs_planted_centres']
        print(k,'b%d'%t['batch'], 'train wall %.1f loop %.1f rate %.0f'%(t['wall_s'],t['loop_s'],t['trainer_loop_patches_per_s']))
        print('  last',{a:round(b,4) for a,b in t['loss_last'].items()})
        print('  det',[round(v,2) for _,v in t['detect_loss_curve']])
        print('  eval wall %.1f loop %.1f dev %.2f'%(e['wall_s'],e['loop_s'],e['device_s']), 'Mpix/s cmd %.1f loop %.1f dev %.1f'%(e['mpix_per_s_whole_command'], e['mpix_per_s_eval_loop'], e['mpix_per_s_network_nms']))
        print('   AP %.3f picks %d'%(m['average_precision'],m['n_picks']), m['at'], m.get('best_f1'))
    if 'pick_agreement_with_fp32' in d:
        print({k:v for k,v in d['pick_agreement_with_fp32'].items() if k!='note'})
        m=d[[k for k in d if k.startswith('fp32_checkpoint')][0]]['picks_vs_planted_centres']
        print('   fp32 ckpt in 16bit: AP %.3f picks %d'%(m['average_precision'],m['n_picks']), m['at'])
