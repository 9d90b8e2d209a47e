int g_dvlp_last_hip_error = 0;
