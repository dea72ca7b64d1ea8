"""ConvLSTM encoder-forecaster — BASELINE config 1 ("ConvLSTM 5->5 frames, batch 2, PyTorch CPU: plumbing, no GPU").

Drop-in for the reference's models/ConvLSTM.py surface: `create_ConvLSTM(output_frames)`, `EF`, `Encoder`, `Forecaster`,
`ConvLSTM` with the reference's constructor arguments and its 46 state_dict keys (ConvLSTM.py:14-32, 100-213, 219-256).
This model is NOT on the north-star hot path: BASELINE names it as the CPU-runnable case that exercises the harness
(trainer loop, loss, fixtures), so it is plain PyTorch ops, no HIP kernels.  Differences that do not change results:

  * peephole weights Wci / Wcf / Wco are registered nn.Parameters on every device (the reference's
    `nn.Parameter(...).to(device)` registers them on CPU and silently drops them on CUDA, ConvLSTM.py:25-27);
  * the module-level singleton encoder / forecaster of the reference (:251-256, shared by every model made by
    create_ConvLSTM) become one fresh pair per call.

Input (B, T_in, 1, 256, 256) -> output (B, T_out, 1, 256, 256); the state grids are 64x64, 16x16, 8x8 (ConvLSTM.py:219-247).
"""
from collections import OrderedDict

import torch
import torch.nn as nn

LEAK = 0.2


class ConvLSTM(nn.Module):
    """One ConvLSTM cell unrolled over `seq_len` steps with peephole terms (ConvLSTM.py:34-69)."""

    def __init__(self, input_channel, num_filter, b_h_w, kernel_size, stride=1, padding=1):
        super().__init__()
        self._conv = nn.Conv2d(input_channel + num_filter, 4 * num_filter, kernel_size, stride, padding)
        self._batch_size, self._state_height, self._state_width = b_h_w
        for n in ("Wci", "Wcf", "Wco"):
            setattr(self, n, nn.Parameter(torch.zeros(1, num_filter, self._state_height, self._state_width)))
        self._input_channel, self._num_filter = input_channel, num_filter

    def forward(self, inputs=None, states=None, seq_len=5):
        ref = inputs if inputs is not None else states[0]
        batch = inputs.size(1) if inputs is not None else states[0].size(0)
        grid = (batch, self._num_filter, self._state_height, self._state_width)
        h, c = states if states is not None else (ref.new_zeros(grid), ref.new_zeros(grid))
        outs = []
        for t in range(seq_len):
            x = inputs[t] if inputs is not None else ref.new_zeros((batch, self._input_channel) + grid[2:])
            gi, gf, gc, go = self._conv(torch.cat((x, h), dim=1)).chunk(4, dim=1)
            i = torch.sigmoid(gi + self.Wci * c)
            f = torch.sigmoid(gf + self.Wcf * c)
            c = f * c + i * torch.tanh(gc)
            o = torch.sigmoid(go + self.Wco * c)    # the output gate peeks at the NEW cell state (:64-66)
            h = o * torch.tanh(c)
            outs.append(h)
        return torch.stack(outs), (h, c)


def make_layers(block):
    """{name: [cin, cout, k, s, p]} -> nn.Sequential; 'deconv' in the name = ConvTranspose2d, 'leaky' = LeakyReLU(0.2)
    after it, 'relu' = ReLU (ConvLSTM.py:72-97)."""
    layers = []
    for name, (cin, cout, k, s, p) in block.items():
        if "deconv" in name:
            layers.append((name, nn.ConvTranspose2d(cin, cout, k, s, p)))
        elif "conv" in name:
            layers.append((name, nn.Conv2d(cin, cout, k, s, p)))
        else:
            raise NotImplementedError(name)
        if "relu" in name:
            layers.append(("relu_" + name, nn.ReLU(inplace=True)))
        elif "leaky" in name:
            layers.append(("leaky_" + name, nn.LeakyReLU(LEAK, inplace=True)))
    return nn.Sequential(OrderedDict(layers))


def _per_frame(net, x):
    s, b = x.shape[:2]
    y = net(x.reshape(s * b, *x.shape[2:]))
    return y.reshape(s, b, *y.shape[1:])


class Encoder(nn.Module):
    def __init__(self, subnets, rnns):
        super().__init__()
        assert len(subnets) == len(rnns)
        self.blocks = len(subnets)
        for i, (params, rnn) in enumerate(zip(subnets, rnns), 1):
            setattr(self, f"stage{i}", make_layers(params))
            setattr(self, f"rnn{i}", rnn)

    def forward(self, x):
        """x: (S, B, C, H, W) -> the (h, c) of every stage (ConvLSTM.py:129-139)."""
        states = []
        for i in range(1, self.blocks + 1):
            x, st = getattr(self, f"rnn{i}")(_per_frame(getattr(self, f"stage{i}"), x), None, seq_len=x.shape[0])
            states.append(st)
        return tuple(states)


class Forecaster(nn.Module):
    def __init__(self, subnets, rnns):
        super().__init__()
        assert len(subnets) == len(rnns)
        self.blocks = len(subnets)
        for i, (params, rnn) in enumerate(zip(subnets, rnns)):
            setattr(self, f"rnn{self.blocks - i}", rnn)
            setattr(self, f"stage{self.blocks - i}", make_layers(params))

    def forward(self, hidden_states, output_seq_len):
        """Deepest state first, no input there (zeros), each stage's output feeds the next (ConvLSTM.py:171-180)."""
        x = None
        for i in range(self.blocks, 0, -1):
            x, _ = getattr(self, f"rnn{i}")(x, hidden_states[i - 1], seq_len=output_seq_len)
            x = _per_frame(getattr(self, f"stage{i}"), x)
        return x


class EF(nn.Module):
    def __init__(self, encoder, forecaster, output_seq_len):
        super().__init__()
        self.encoder, self.forecaster, self.output_seq_len = encoder, forecaster, output_seq_len

    def forward(self, x):
        """(B, S, C, H, W) -> (B, S_out, C, H, W) (ConvLSTM.py:192-197)."""
        out = self.forecaster(self.encoder(x.permute(1, 0, 2, 3, 4)), self.output_seq_len)
        return out.permute(1, 0, 2, 3, 4)


batch_size = 4


def _cells(spec):
    return [ConvLSTM(input_channel=c, num_filter=f, b_h_w=(batch_size, g, g), kernel_size=3, stride=1, padding=1) for c, f, g in spec]


def create_ConvLSTM(output_frames):
    """The reference's configuration (ConvLSTM.py:214-247): 256x256 -> 64x64x8 -> 16x16x192 -> 8x8x192 and back."""
    enc_nets = [{"conv1_leaky_1": [1, 8, 6, 4, 1]}, {"conv2_leaky_1": [64, 192, 4, 4, 1]}, {"conv3_leaky_1": [192, 192, 3, 2, 1]}]
    dec_nets = [{"deconv1_leaky_1": [192, 192, 4, 2, 1]}, {"deconv2_leaky_1": [192, 64, 6, 4, 1]},
                {"deconv3_leaky_1": [64, 8, 6, 4, 1], "conv3_leaky_2": [8, 8, 3, 1, 1], "conv3_3": [8, 1, 1, 1, 0]}]
    encoder = Encoder(enc_nets, _cells([(8, 64, 64), (192, 192, 16), (192, 192, 8)]))
    forecaster = Forecaster(dec_nets, _cells([(192, 192, 8), (192, 192, 16), (64, 64, 64)]))
    return EF(encoder, forecaster, output_frames)
