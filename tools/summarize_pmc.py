#!/usr/bin/env python
"""Condense rocprofv3 counter_collection / kernel_stats CSVs to our own kernels (short names) for profiles/."""
import csv, re, sys, collections

def short(name):
    m = re.search(r"(gemm_f16_w4_kernel|gemm_f16_p8_kernel|gemm_f16_p4_kernel|gemm_f16_ring_kernel|gemm_f16_kernel|gather_rows_kernel|attend_kernel|attend_hidden_kernel|attend_hidden_bwd_kernel|gather_rows_bwd_kernel|gather_bbox_kernel|local_hidden_kernel|local_mlp_kernel|row_stats_kernel|col_stats_kernel|dual_softmax_apply_kernel|dual_softmax_bwd_apply_kernel|conv_wgrad_planes_kernel|conv_wgrad_reduce_kernel|dwconv3x3_wgrad_kernel|gn_relu_bwd_reduce_kernel|gn_relu_bwd_apply_kernel|conv4d_k3s1_kernel|resize_bilinear_ac_kernel|"
                  r"linear_f32_kernel|conv4d_kernel|gn_relu_kernel|gemm_nt_f32_kernel|l2norm_rows_kernel|soft_argmax_rows_kernel|soft_argmax_cols_kernel|sample_geometry_kernel|project_rays_kernel|nchw_to_nhwc_f16_kernel|"
                  r"mask_rgb_kernel|pack_weight_f16_kernel|ray_mlp_kernel|fused_\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else None

def main(path, out):
    rows = list(csv.DictReader(open(path)))
    if rows and "Counter_Name" in rows[0]:
        agg = collections.OrderedDict()
        for r in rows:
            k = short(r["Kernel_Name"])
            if k:
                agg.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
        with open(out, "w") as f:
            f.write("kernel,counter,mean_per_dispatch,dispatches\n")
            for (k, c), v in agg.items():
                f.write(f'"{k}",{c},{sum(v)/len(v):.3f},{len(v)}\n')
    else:
        tot = sum(int(r["TotalDurationNs"]) for r in rows)
        with open(out, "w") as f:
            f.write("kernel,calls,total_ms,avg_us,pct_of_all_gpu_time\n")
            other = 0
            for r in rows:
                k = short(r["Name"])
                if k:
                    f.write(f'"{k}",{r["Calls"]},{int(r["TotalDurationNs"])/1e6:.3f},{float(r["AverageNs"])/1e3:.2f},'
                            f'{100.0*int(r["TotalDurationNs"])/tot:.2f}\n')
                else:
                    other += int(r["TotalDurationNs"])
            f.write(f'"(torch plumbing: aux outputs, copies, 4x4 algebra)",,{other/1e6:.3f},,{100.0*other/tot:.2f}\n')

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
