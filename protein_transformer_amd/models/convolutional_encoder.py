"""`-m conv-enc`: 1-D sequence convolutions between the embedding and the encoder layers.

Drop-in for /root/reference/protein_transformer/models/convolutional_encoder.py:13-129
(`ConvEncoderOnlyTransformer(nlayers, nhead, dmodel, dff, max_seq_len, vocab, angle_means, use_tanh_out,
conv_kernel_sizes, conv_dim_reductions, use_embedding, conv_out_matches_dm, dropout)`), same `state_dict` keys
(`encoder.conv_layers.j.{weight [Cout,Cin,k], bias}`).  The convolutions have no activation in between
(:113-114), preserve the length (odd kernels, zero padding inside each protein), and run as im2col + the
fp32 MFMA GEMM (csrc/conv.hip).  With no kernel sizes the model is exactly `enc-only`, as upstream.
"""
from .encoder_only import _TransformerBase


class ConvEncoderOnlyTransformer(_TransformerBase):
    """ A Transformer that starts with 1D sequence convolutions before applying attention. """

    def __init__(self, nlayers, nhead, dmodel, dff, max_seq_len, vocab, angle_means, use_tanh_out, conv_kernel_sizes,
                 conv_dim_reductions, use_embedding, conv_out_matches_dm, dropout=0.1):
        super().__init__(nlayers, nhead, dmodel, dff, max_seq_len, vocab, angle_means, use_tanh_out, dropout=dropout,
                         conv_kernel_sizes=list(conv_kernel_sizes), conv_dim_reductions=list(conv_dim_reductions),
                         use_embedding=use_embedding, conv_out_matches_dm=conv_out_matches_dm)
