"""G-data fixtures: ml-100k through the REAL reference TripletDataset / SeqDataset.

Records the reference's outputs (ids, splits, histories, first batches) plus the raw interaction
columns of its bundled demo file (recstudio/dataset_demo/ml-100k, a data file the reference ships)
in compact integer form, so that the product loaders can be checked on any machine."""
import os

import numpy as np
import torch


def gen_data(out_dir):
    import pandas as pd
    from recstudio.data.dataset import TripletDataset, SeqDataset
    from recstudio.utils import seed_everything, get_dataset_default_config
    demo = '/root/reference/recstudio/dataset_demo/ml-100k'
    raw = pd.read_csv(os.path.join(demo, 'ml-100k.inter'), sep='\t', header=0)
    users_file = pd.read_csv(os.path.join(demo, 'ml-100k.user'), sep='\t', header=0)
    out = {'raw_user': raw.iloc[:, 0].values.astype(np.int32), 'raw_item': raw.iloc[:, 1].values.astype(np.int32),
           'raw_rating': raw.iloc[:, 2].values.astype(np.int8), 'raw_time': raw.iloc[:, 3].values.astype(np.int32),
           'user_file_ids': users_file.iloc[:, 0].values.astype(np.int32)}

    def np_(t):
        return t.detach().cpu().numpy()

    # --- TripletDataset, quickstart order: seed first, then build (quickstart/run.py:35,56)
    seed_everything(2022)
    ds = TripletDataset('ml-100k')
    trn, val, tst = ds.build(split_ratio=[0.8, 0.1, 0.1], shuffle=True, split_mode='user_entry')
    out.update(t_num_users=ds.num_users, t_num_items=ds.num_items, t_num_inter=len(ds.inter_feat),
               t_train_index=np_(trn.data_index).astype(np.int32), t_val_index=np_(val.data_index).astype(np.int32),
               t_test_index=np_(tst.data_index).astype(np.int32),
               t_inter_user=np_(trn.inter_feat.get_col('user_id')).astype(np.int16),
               t_inter_item=np_(trn.inter_feat.get_col('item_id')).astype(np.int16),
               t_inter_rating=np_(trn.inter_feat.get_col('rating')).astype(np.int8),
               t_item_freq=np_(trn.item_freq).astype(np.int32), t_user_count=np_(trn.user_count).astype(np.int32),
               t_user_hist_shape=np.array(trn.user_hist.shape), t_user_hist_head=np_(trn.user_hist[:8]).astype(np.int16),
               t_test_user_hist_shape=np.array(tst.user_hist.shape),
               t_test_user_hist_head=np_(tst.user_hist[:8]).astype(np.int16),
               t_test_user_count=np_(tst.user_count).astype(np.int32))
    trn.use_field = {'user_id', 'item_id', 'rating'}
    trn.drop_feat(trn.use_field)
    val.use_field = trn.use_field
    for i, b in enumerate(trn.train_loader(batch_size=512, shuffle=False)):
        if i >= 2:
            break
        assert sorted(b.keys()) == ['item_id', 'rating', 'user_id'], b.keys()
        out[f't_trainbatch{i}_user'], out[f't_trainbatch{i}_item'] = np_(b['user_id']), np_(b['item_id'])
        out[f't_trainbatch{i}_rating'] = np_(b['rating'])
    for i, b in enumerate(val.eval_loader(batch_size=20)):
        if i >= 2:
            break
        for k in ('user_id', 'item_id', 'rating', 'user_hist'):
            out[f't_valbatch{i}_{k}'] = np_(b[k])
    out['t_n_train_batches'] = len(trn.train_loader(batch_size=512, shuffle=False))
    out['t_n_val_batches'] = len(val.eval_loader(batch_size=20))

    # --- SeqDataset (SASRec's dataset class), max_seq_len = 50
    seed_everything(2022)
    cfg = {'max_seq_len': 50}
    sq = SeqDataset('ml-100k', cfg)
    strn, sval, stst = sq.build(split_ratio=2, split_mode='user_entry')
    out.update(s_num_users=sq.num_users, s_num_items=sq.num_items, s_num_inter=len(sq.inter_feat),
               s_sizes=np.array([len(strn), len(sval), len(stst)]),
               s_train_index_head=np_(strn.data_index[:300]).astype(np.int32),
               s_train_index_tail=np_(strn.data_index[-300:]).astype(np.int32),
               s_val_index=np_(sval.data_index).astype(np.int32), s_test_index=np_(stst.data_index).astype(np.int32),
               s_inter_item=np_(strn.inter_feat.get_col('item_id')).astype(np.int16),
               s_item_freq=np_(strn.item_freq).astype(np.int32))
    strn.use_field = {'user_id', 'item_id', 'rating'}
    strn.drop_feat(strn.use_field)
    for i, b in enumerate(strn.train_loader(batch_size=64, shuffle=False)):
        if i >= 2:
            break
        out[f's_trainbatch{i}_keys'] = np.array(sorted(b.keys()))
        for k in ('user_id', 'seqlen', 'in_item_id', 'item_id', 'in_rating', 'rating'):
            out[f's_trainbatch{i}_{k}'] = np_(b[k])
    np.savez_compressed(os.path.join(out_dir, 'data_ml100k.npz'), **out)
