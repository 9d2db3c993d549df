# same-box A/B of the round-5 small-batch changes (modes: name:ENV=V;ENV=V,...)
export DSP_R5_MODES="${DSP_R5_MODES:-round4:DSP_LSTM_FRONT_CLUSTER=0;DSP_FC_SMALL=0;DSP_LSTM_HANDOFF=0,fc_small:DSP_LSTM_FRONT_CLUSTER=0;DSP_LSTM_HANDOFF=0,front:DSP_LSTM_HANDOFF=0,auto:}"
timeout 900 python tools/experiments/r5_small_ab.py ${1:-512,1024,2048,4096} ${2:-300}
